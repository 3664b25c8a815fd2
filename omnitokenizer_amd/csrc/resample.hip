// Token-grid resampling for the optional pooling blocks and the deferred pools of the
// 'linear' patch-embed.  Pure HBM streaming: one float4 of one output token per thread.
//   avg2d / max2d  Pooling('a' / 'm'), reference attention.py:83-106 (AvgPool2d(2) / MaxPool2d(2) over
//                  the sqrt(N) x sqrt(N) token grid of each (b t) row group)
//   up2d           nn.Upsample(scale_factor=(1,2,2), 'nearest'), omnitokenizer.py:1001
//   avg_t          AvgPool3d((2,1,1)) over the frames after the first, omnitokenizer.py:909-914
//   up_t           nn.Upsample(scale_factor=(2,1,1), 'nearest') over the frames after the first,
//                  omnitokenizer.py:1103-1107
#include "common.h"

namespace omnitok {

enum { RS_AVG2D = 0, RS_MAX2D = 1, RS_UP2D = 2, RS_AVG_T = 3, RS_UP_T = 4 };

template <int MODE>
__global__ __launch_bounds__(256) void token_resample_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ y,
                                                             int64_t n, int T, int gh, int gw, int d4) {
    // output geometry
    int To = 1, oh = gh, ow = gw;
    if (MODE == RS_AVG2D || MODE == RS_MAX2D) { oh = gh / 2; ow = gw / 2; }
    if (MODE == RS_UP2D) { oh = gh * 2; ow = gw * 2; }
    if (MODE == RS_AVG_T) To = 1 + (T - 1) / 2;
    if (MODE == RS_UP_T) To = 1 + (T - 1) * 2;
    const bool temporal = MODE == RS_AVG_T || MODE == RS_UP_T;
    const int64_t total = n * (temporal ? (int64_t)To : 1) * oh * ow * d4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int d = (int)(idx % d4);
    int64_t r = idx / d4;
    if (temporal) {
        const int64_t S = (int64_t)gh * gw;
        const int64_t s = r % S; r /= S;
        const int to = (int)(r % To);
        const int64_t b = r / To;
        const f32x4 *xb = x + (b * T * S + s) * d4 + d;
        if (MODE == RS_AVG_T) {
            if (to == 0) {
                y[idx] = xb[0];
            } else {
                const f32x4 a = xb[(int64_t)(2 * to - 1) * S * d4], c = xb[(int64_t)(2 * to) * S * d4];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (a[e] + c[e]) * 0.5f;  // sum / 2, exact scaling
                y[idx] = o;
            }
        } else {
            const int ti = to == 0 ? 0 : 1 + (to - 1) / 2;
            y[idx] = xb[(int64_t)ti * S * d4];
        }
        return;
    }
    const int xo = (int)(r % ow); r /= ow;
    const int yo = (int)(r % oh);
    const int64_t b = r / oh;
    const f32x4 *xb = x + b * gh * gw * d4 + d;
    if (MODE == RS_UP2D) {
        y[idx] = xb[((int64_t)(yo / 2) * gw + xo / 2) * d4];
        return;
    }
    // 2x2 window in the row-major order ATen's pooling loops use: (0,0) (0,1) (1,0) (1,1)
    const f32x4 a = xb[((int64_t)(2 * yo) * gw + 2 * xo) * d4];
    const f32x4 c = xb[((int64_t)(2 * yo) * gw + 2 * xo + 1) * d4];
    const f32x4 g = xb[((int64_t)(2 * yo + 1) * gw + 2 * xo) * d4];
    const f32x4 h = xb[((int64_t)(2 * yo + 1) * gw + 2 * xo + 1) * d4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (MODE == RS_AVG2D)
            o[e] = (((a[e] + c[e]) + g[e]) + h[e]) * 0.25f;
        else
            o[e] = fmaxf(fmaxf(a[e], c[e]), fmaxf(g[e], h[e]));
    }
    y[idx] = o;
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_token_resample(const float *x, float *out, int mode, int64_t n, int T, int gh, int gw, int D,
                                      omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(mode >= 0 && mode <= 4, "token_resample: mode %d", mode);
    OT_CHECK_ARG(D % 4 == 0 && gh > 0 && gw > 0 && T >= 1, "token_resample: bad shape");
    if (mode == RS_AVG2D || mode == RS_MAX2D)
        OT_CHECK_ARG(gh % 2 == 0 && gw % 2 == 0, "token_resample: %dx%d grid is not even", gh, gw);
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(x && out, "token_resample: null pointer");
    int64_t tokens = n * gh * gw;
    switch (mode) {
        case RS_AVG2D: case RS_MAX2D: tokens /= 4; break;
        case RS_UP2D: tokens *= 4; break;
        case RS_AVG_T: tokens *= 1 + (T - 1) / 2; break;
        case RS_UP_T: tokens *= 1 + (T - 1) * 2; break;
    }
    const int d4 = D / 4;
    const dim3 grid((unsigned)((tokens * d4 + 255) / 256));
    const f32x4 *xi = reinterpret_cast<const f32x4 *>(x);
    f32x4 *yo = reinterpret_cast<f32x4 *>(out);
#define RS_LAUNCH(M) hipLaunchKernelGGL(token_resample_kernel<M>, grid, dim3(256), 0, stream, xi, yo, n, T, gh, gw, d4)
    switch (mode) {
        case RS_AVG2D: RS_LAUNCH(RS_AVG2D); break;
        case RS_MAX2D: RS_LAUNCH(RS_MAX2D); break;
        case RS_UP2D: RS_LAUNCH(RS_UP2D); break;
        case RS_AVG_T: RS_LAUNCH(RS_AVG_T); break;
        default: RS_LAUNCH(RS_UP_T); break;
    }
#undef RS_LAUNCH
    OT_LAUNCH_CHECK("token_resample");
    return OMNITOK_OK;
}
