// PEG: depthwise 3x3x3 convolution over the token grid as a channel-last stencil, with bias and
// the caller's residual fused (reference attention.py:298-338 + :667).  HBM-bound: 27 MAC/element,
// so the kernel is organised for 16-byte channel-contiguous accesses and re-uses each loaded
// input vector for up to three outputs along w (4 outputs per thread -> 54 instead of 108 loads).
//
// The input is the raw contiguous token buffer viewed as [B,T,H,W,D] -- for the temporal
// transformers this is the reference's "scrambled" view of a '(b h w) t d' buffer
// (attention.py:319); the engine keeps tokens physically in the layout the
// reference has at that point, so the same kernel reproduces both cases.
#include "common.h"
#include "peg_wide.h"

namespace omnitok {

constexpr int PEG_WSEG = 4;
int g_peg_variant = 1;  // omnitok_set_option("peg_variant"): 0 register-blocked | 1 LDS-tiled: the 64-channel kernel of peg_wide.h for 2..8 planes
                        // on grids it accepts, else the time ring below | 2 peg_wide.h whenever the grid allows | 3 time ring only

__global__ __launch_bounds__(256) void peg3d_kernel(const float *__restrict__ x, const float *__restrict__ w27,
                                                    const float *__restrict__ bias, float *__restrict__ y, int B,
                                                    int T, int H, int W, int D, int tpad) {
    const int d4n = D >> 2;
    const int wsegs = (W + PEG_WSEG - 1) / PEG_WSEG;
    // thread -> (position segment, channel quad); consecutive threads = consecutive channels
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * T * H * wsegs * d4n;
    if (gid >= total) return;
    const int c4 = (int)(gid % d4n);
    int64_t rr = gid / d4n;
    const int ws = (int)(rr % wsegs); rr /= wsegs;
    const int h = (int)(rr % H); rr /= H;
    const int t = (int)(rr % T); rr /= T;
    const int b = (int)rr;
    const int w0 = ws * PEG_WSEG;

    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    const f32x4 *wt4 = reinterpret_cast<const f32x4 *>(w27);
    const f32x4 bv = reinterpret_cast<const f32x4 *>(bias)[c4];
    f32x4 acc[PEG_WSEG];
#pragma unroll
    for (int j = 0; j < PEG_WSEG; ++j) acc[j] = bv;

#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int ti = t + dt - tpad;
        if (ti < 0 || ti >= T) continue;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = h + dh - 1;
            if (hi < 0 || hi >= H) continue;
            const int64_t rowbase = (((int64_t)b * T + ti) * H + hi) * W;
            f32x4 in[PEG_WSEG + 2];
#pragma unroll
            for (int j = 0; j < PEG_WSEG + 2; ++j) {
                const int wi = w0 + j - 1;
                if (wi >= 0 && wi < W)
                    in[j] = x4[(rowbase + wi) * d4n + c4];
                else
                    in[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const f32x4 k0 = wt4[((dt * 3 + dh) * 3 + 0) * d4n + c4];
            const f32x4 k1 = wt4[((dt * 3 + dh) * 3 + 1) * d4n + c4];
            const f32x4 k2 = wt4[((dt * 3 + dh) * 3 + 2) * d4n + c4];
#pragma unroll
            for (int j = 0; j < PEG_WSEG; ++j) acc[j] += in[j] * k0 + in[j + 1] * k1 + in[j + 2] * k2;
        }
    }
    const int64_t obase = (((int64_t)b * T + t) * H + h) * W;
#pragma unroll
    for (int j = 0; j < PEG_WSEG; ++j) {
        const int wo = w0 + j;
        if (wo < W) {
            const int64_t o = (obase + wo) * d4n + c4;
            reinterpret_cast<f32x4 *>(y)[o] = acc[j] + x4[o];  // + residual
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled variant (default when D % 32 == 0).  The register-blocked kernel above re-reads every
// input vector ~13 times through L1 and measured 2.1 TB/s effective at C3 (28 % of HBM peak).
// Here a workgroup owns a 4 x 32 spatial tile of 32 channels (128 B per position = one cache
// line) and walks the time axis with a 3-plane ring in LDS: every input plane tile (with its
// one-row/one-column halo) is fetched from HBM once per workgroup, the 27 weights of the thread's
// channel quad stay in registers, the residual comes from the centre tap in LDS, and the next
// plane's global loads are in flight while the current output plane is computed.
// HBM traffic per call: reads 1.6x (row halo) + writes 1x of the token tensor.
// ---------------------------------------------------------------------------------------------
constexpr int PT_H = 4, PT_W = 32, PT_C4 = 8;                   // tile rows, cols, float4 channels
constexpr int PT_PLANE_V4 = (PT_H + 2) * (PT_W + 2) * PT_C4;    // 1632 float4 per plane tile
constexpr int PT_LDS_BYTES = 3 * PT_PLANE_V4 * 16;              // 78336
constexpr int PT_LOADS = (PT_PLANE_V4 + 255) / 256;             // 7 float4 per thread per plane

__global__ __launch_bounds__(256, 2) void peg3d_lds_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ w27,
                                                            const float *__restrict__ bias, float *__restrict__ y,
                                                            int B, int T, int H, int W, int D, int tpad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4 *ring = reinterpret_cast<f32x4 *>(smem);
    const int tid = threadIdx.x;
    const int c4 = tid & 7, wcol = tid >> 3;  // 8 channel quads x 32 columns
    const int wtiles = (W + PT_W - 1) / PT_W, htiles = (H + PT_H - 1) / PT_H;
    // XCD-aware order (workgroup n runs on XCD n % 8, each XCD has its own L2): the spatial tiles of ONE (clip, channel
    // slab) -- which share their halo rows -- are consecutive workgroups of the SAME XCD, so a halo row is fetched from
    // HBM once instead of once per neighbouring tile (counted reads were 1.45x the tensor with tiles dealt round-robin).
    const int per = wtiles * htiles, ngroups = gridDim.x / per;  // groups = (clip, slab) pairs
    const int n = blockIdx.x;
    int grp, tile;
    {
        const int xcd = n & 7, j = n >> 3;
        const int full = (ngroups / 8) * 8;  // groups dealt in whole rounds of 8; the remainder falls back to linear order
        const int g = xcd + 8 * (j / per);
        if (g < full) {
            grp = g;
            tile = j % per;
        } else {
            const int rest = n - full * per;  // workgroups after the whole rounds (n >= full * per here)
            grp = full + rest / per;
            tile = rest % per;
        }
    }
    const int w0 = (tile % wtiles) * PT_W, h0 = (tile / wtiles) * PT_H;
    const int nslab = D >> 5;
    const int slab = grp % nslab, b = grp / nslab;
    const int d4n = D >> 2;
    const int ch4 = slab * PT_C4 + c4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);

    // the 27 taps of this thread's channel quad
    f32x4 wt[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wt[k] = reinterpret_cast<const f32x4 *>(w27)[k * d4n + ch4];
    const f32x4 bv = reinterpret_cast<const f32x4 *>(bias)[ch4];

    // loader: element i -> (position pos = i / 8, channel quad i % 8); pos -> (row r, col c) of the halo tile
    f32x4 stage[PT_LOADS];
    auto gload = [&](int tau) {
#pragma unroll
        for (int k = 0; k < PT_LOADS; ++k) {
            const int i = tid + 256 * k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < PT_PLANE_V4) {
                const int pos = i >> 3, q = i & 7;
                const int r = pos / (PT_W + 2), c = pos - r * (PT_W + 2);
                const int hh = h0 - 1 + r, ww = w0 - 1 + c;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W)
                    v = x4[((((int64_t)b * T + tau) * H + hh) * W + ww) * d4n + slab * PT_C4 + q];
            }
            stage[k] = v;
        }
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int k = 0; k < PT_LOADS; ++k) {
            const int i = tid + 256 * k;
            if (i < PT_PLANE_V4) ring[slot * PT_PLANE_V4 + i] = stage[k];
        }
    };

    const int n_iter = T + 2 - tpad;  // planes 0..T-1, plus one empty step for the non-causal case
    gload(0);
    for (int tau = 0; tau < n_iter; ++tau) {
        if (tau < T) lstore(tau % 3);
        __syncthreads();
        if (tau + 1 < T) gload(tau + 1);  // in flight while this step computes
        const int t = tau - 2 + tpad;     // output plane completed by input plane tau
        if (t >= 0 && t < T && w0 + wcol < W) {
            f32x4 acc[PT_H];
#pragma unroll
            for (int r = 0; r < PT_H; ++r) acc[r] = bv;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int ti = t + dt - tpad;
                if (ti < 0 || ti >= T) continue;
                const f32x4 *pl = ring + (ti % 3) * PT_PLANE_V4;
#pragma unroll
                for (int r = 0; r < PT_H + 2; ++r)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const f32x4 v = pl[(r * (PT_W + 2) + wcol + dw) * PT_C4 + c4];
#pragma unroll
                        for (int dh = 0; dh < 3; ++dh) {
                            const int orow = r - dh;  // input row r feeds output row r - dh with tap dh
                            if (orow >= 0 && orow < PT_H) acc[orow] += v * wt[(dt * 3 + dh) * 3 + dw];
                        }
                    }
            }
            // residual: centre tap of plane t
            const f32x4 *pc = ring + (t % 3) * PT_PLANE_V4;
#pragma unroll
            for (int r = 0; r < PT_H; ++r) {
                const int hh = h0 + r;
                if (hh < H) {
                    const f32x4 xc = pc[((r + 1) * (PT_W + 2) + wcol + 1) * PT_C4 + c4];
                    reinterpret_cast<f32x4 *>(y)[((((int64_t)b * T + t) * H + hh) * W + w0 + wcol) * d4n + ch4] =
                        acc[r] + xc;
                }
            }
        }
        __syncthreads();  // everyone is done with slot (tau+1) % 3 == (tau-2) % 3 before it is refilled
    }
}

__global__ void pack_peg_kernel(const float *__restrict__ w, int D, float *__restrict__ w27) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 27 * D) return;
    const int tap = idx / D, c = idx % D;
    w27[idx] = w[c * 27 + tap];  // [D,1,3,3,3] -> [27, D]
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_peg3d(const float *x, const float *w27, const float *bias, float *y, int B, int T, int H,
                             int W, int D, int causal, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && w27 && bias && y && x != y, "peg3d: bad pointers (y must not alias x)");
    OT_CHECK_ARG(D % 4 == 0, "peg3d: D %% 4 != 0");
    OT_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(w27) && aligned16(bias), "peg3d: unaligned pointer");
    if (B * T * H * W == 0) return OMNITOK_OK;
    // 64-channel slab: 164 vs 188 us at C3's [32, 5, 32, 32] (tools/probes/peg_wide_probe.hip, bit-identical); no gain at one plane
    // (221 vs 224 us) and a loss on long walks over large grids (17 planes of 64 x 64: 68 vs 62 us) -- hence the plane-count window
    const bool wide_ok = D % 64 == 0 && W % pegw::W_T == 0 && H % pegw::H_T == 0 &&
                         (int64_t)(W / pegw::W_T) * (H / pegw::H_T) * (D / 64) * B < (1ll << 31);
    if (wide_ok && T == 1 && (g_peg_variant == 1 || g_peg_variant == 4)) {
        // images: the one-plane kernel (9 taps, four workgroups per CU); "peg_variant" 2 keeps the walk kernel for the A/B
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(pegw::peg2d_wide_kernel), pegw::LDS_BYTES_2D)) return rc;
        hipLaunchKernelGGL(pegw::peg2d_wide_kernel, dim3((unsigned)((int64_t)(W / pegw::W_T) * (H / pegw::H_T) * (D / 64) * B)), dim3(256),
                           pegw::LDS_BYTES_2D, stream, x, w27, bias, y, B, H, W, D, causal ? 2 : 1);
        OT_LAUNCH_CHECK("peg2d_wide");
        return OMNITOK_OK;
    }
    if (wide_ok && ((g_peg_variant == 1 && T >= 2 && T <= 8) || g_peg_variant == 2)) {
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(pegw::peg3d_wide_kernel), pegw::LDS_BYTES)) return rc;
        hipLaunchKernelGGL(pegw::peg3d_wide_kernel, dim3((unsigned)((int64_t)(W / pegw::W_T) * (H / pegw::H_T) * (D / 64) * B)), dim3(256),
                           pegw::LDS_BYTES, stream, x, w27, bias, y, B, T, H, W, D, causal ? 2 : 1);
        OT_LAUNCH_CHECK("peg3d_wide");
        return OMNITOK_OK;
    }
    if (D % 32 == 0 && g_peg_variant >= 1 && (int64_t)((W + PT_W - 1) / PT_W) * ((H + PT_H - 1) / PT_H) * (D / 32) * B < (1ll << 31)) {
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(peg3d_lds_kernel), PT_LDS_BYTES)) return rc;
        const int wtiles = (W + PT_W - 1) / PT_W, htiles = (H + PT_H - 1) / PT_H;
        hipLaunchKernelGGL(peg3d_lds_kernel, dim3((unsigned)((int64_t)wtiles * htiles * (D / 32) * B)), dim3(256), PT_LDS_BYTES,
                           stream, x, w27, bias, y, B, T, H, W, D, causal ? 2 : 1);
        OT_LAUNCH_CHECK("peg3d_lds");
        return OMNITOK_OK;
    }
    const int wsegs = (W + PEG_WSEG - 1) / PEG_WSEG;
    const int64_t total = (int64_t)B * T * H * wsegs * (D / 4);
    hipLaunchKernelGGL(peg3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, w27, bias, y, B,
                       T, H, W, D, causal ? 2 : 1);
    OT_LAUNCH_CHECK("peg3d");
    return OMNITOK_OK;
}

extern "C" int omnitok_pack_peg_weight(const float *w, int D, float *w27, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(pack_peg_kernel, dim3((27 * D + 255) / 256), dim3(256), 0, stream, w, D, w27);
    OT_LAUNCH_CHECK("pack_peg");
    return OMNITOK_OK;
}
