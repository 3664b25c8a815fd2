// HBM-bound row kernels: LayerNorm, patch gather + LayerNorm, un-patchify scatter and the
// token-order transpose.  One wave64 per row, 16-byte accesses, no LDS.
#include "common.h"
#include "planes.h"

namespace omnitok {

constexpr int LN_MAX_V4 = 4;  // up to 4 float4 per lane -> dim <= 1024

// mean / rstd of a row held as nv float4 per lane (two-pass: mean, then centred sum of squares)
// Every operation is stated (contraction off, the fused multiply-adds written out): left to the compiler, the two squares
// of a pair were fused as fma(a, a, b * b) in one kernel and fma(b, b, a * a) in another that inlines this same function
// with compile-time sizes -- 6 % of rows then differed by one unit in the last place of rstd between layernorm_kernel and
// layernorm_prevq_kernel (r05, tools/r05/prevq_debug.py).  Now every kernel that shares this function rounds alike.
__device__ __forceinline__ void row_stats(const f32x4 *v, int nv, int lane, int dim, float eps, float &mean,
                                          float &rstd) {
#pragma clang fp contract(off)
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (i < nv && (lane + 64 * i) * 4 < dim) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    s = wave_allsum(s);
    mean = s / (float)dim;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (i < nv && (lane + 64 * i) * 4 < dim) {
            const float a = v[i][0] - mean, b = v[i][1] - mean, c = v[i][2] - mean, d = v[i][3] - mean;
            q += __builtin_fmaf(a, a, b * b) + __builtin_fmaf(c, c, d * d);
        }
    q = wave_allsum(q);
    rstd = 1.0f / sqrtf(q / (float)dim + eps);
}

// (v - mean) * rstd * gamma as three separately rounded operations: what layernorm_kernel has always computed (the bias add
// sits in its own branch), stated explicitly so that every kernel sharing it rounds the same way
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float g) {
#pragma clang fp contract(off)
    return (v - mean) * rstd * g;
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float *__restrict__ y,
                                                        int64_t rows, int dim, float eps, int64_t rpg,
                                                        int64_t gstride, int64_t goff, int tr_a, int tr_c) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = (dim / 4 + 63) / 64;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + row * dim);
    f32x4 v[LN_MAX_V4];
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (i < nv && (lane + 64 * i) * 4 < dim) v[i] = xr[lane + 64 * i];
    float mean, rstd;
    row_stats(v, nv, lane, dim, eps, mean, rstd);
    int64_t orow = row;
    if (rpg > 0) orow = (row / rpg) * gstride + goff + (row % rpg);
    if (tr_a > 0) {  // token transpose fused into the store: row (b, a, c) -> (b, c, a)
        const int64_t c = row % tr_c, ba = row / tr_c;
        orow = ((ba / tr_a) * tr_c + c) * tr_a + ba % tr_a;
    }
    f32x4 *yr = reinterpret_cast<f32x4 *>(y + orow * dim);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gamma);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(beta);
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (i < nv && (lane + 64 * i) * 4 < dim) {
            const f32x4 g = g4[lane + 64 * i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ln_apply(v[i][e], mean, rstd, g[e]);
            if (beta) {
                const f32x4 b = b4[lane + 64 * i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += b[e];
            }
            yr[lane + 64 * i] = o;
        }
}

// The encoder's last norm_out LayerNorm with pre_vq (Linear dim -> 8, + bias, F.normalize; reference omnitokenizer.py:143-148,
// 251-252) applied to the normalised row while it is still on chip: the row goes through a wave-private LDS block instead of
// a 2 KiB HBM write + read per token.  Arithmetic = layernorm_kernel followed by pre_vq_kernel (vq.hip), operation for
// operation (same lane -> element maps, same reduction trees, the shared prevq_* helpers of common.h), so z is bit-identical
// to the two-pass flow (tests/test_gpu_ops.py::test_layernorm_prevq_matches_two_pass).  A wave walks PV_ROWS consecutive rows
// with the next row's loads in flight; the four 16-lane DPP rows of the wave take two of the eight output channels each and
// keep their slice of the weight (2 x dim / 16 floats per lane) in registers for the whole walk.  No workgroup barrier: the
// LDS block is private to the wave, whose LDS instructions execute in order.  z row = the LayerNorm's output row.
constexpr int PV_ROWS = 8;
template <int NV>  // float4 per lane of one row: dim = 256 * NV
__global__ __launch_bounds__(256) void layernorm_prevq_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, const float *__restrict__ w,
                                                             const float *__restrict__ b, float *__restrict__ z, int64_t rows,
                                                             float eps, int tr_a, int tr_c, int l2) {
    constexpr int dim = 256 * NV;
    __shared__ __attribute__((aligned(16))) float rowbuf[4][dim];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, grp = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * PV_ROWS;
    if (row0 >= rows) return;
    const int64_t nrow = rows - row0 < PV_ROWS ? rows - row0 : PV_ROWS;
    // this DPP row's two channels of the weight, in pre_vq_kernel's lane -> element map (elements l16 * 4 + 64 j)
    f32x4 wr[2][4 * NV];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 4 * NV; ++j) wr[cc][j] = *reinterpret_cast<const f32x4 *>(w + (2 * grp + cc) * dim + l16 * 4 + 64 * j);
    const float bias0 = b[2 * grp], bias1 = b[2 * grp + 1];
    f32x4 g4[NV], b4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g4[i] = reinterpret_cast<const f32x4 *>(gamma)[lane + 64 * i];
        b4[i] = beta ? reinterpret_cast<const f32x4 *>(beta)[lane + 64 * i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 v[LN_MAX_V4], vn[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) vn[i] = reinterpret_cast<const f32x4 *>(x + row0 * dim)[lane + 64 * i];
    for (int64_t r = 0; r < nrow; ++r) {
        const int64_t row = row0 + r;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = vn[i];
        {
            const int64_t nx = r + 1 < nrow ? row + 1 : row;  // clamped: the last prefetch re-reads the row (branch-free)
#pragma unroll
            for (int i = 0; i < NV; ++i) vn[i] = reinterpret_cast<const f32x4 *>(x + nx * dim)[lane + 64 * i];
        }
        float mean, rstd;
        row_stats(v, NV, lane, dim, eps, mean, rstd);
        int64_t orow = row;
        if (tr_a > 0) {
            const int64_t c = row % tr_c, ba = row / tr_c;
            orow = ((ba / tr_a) * tr_c + c) * tr_a + ba % tr_a;
        }
        f32x4 *yr = reinterpret_cast<f32x4 *>(rowbuf[wave]);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ln_apply(v[i][e], mean, rstd, g4[i][e]);
            if (beta) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += b4[i][e];
            }
            yr[lane + 64 * i] = o;
        }
        __builtin_amdgcn_wave_barrier();
        float acc[2] = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4 * NV; ++j) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(rowbuf[wave] + l16 * 4 + 64 * j);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                acc[cc] += prevq_dot4(xv[0], xv[1], xv[2], xv[3], wr[cc][j][0], wr[cc][j][1], wr[cc][j][2], wr[cc][j][3]);
        }
        __builtin_amdgcn_wave_barrier();
        acc[0] = row16_allsum(acc[0]) + bias0;
        acc[1] = row16_allsum(acc[1]) + bias1;
        float zc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) zc[c] = __shfl(acc[c & 1], (c >> 1) * 16);
        if (l2) prevq_l2norm8(zc);
        if (lane == 0) *reinterpret_cast<f32x4 *>(z + orow * 8) = f32x4{zc[0], zc[1], zc[2], zc[3]};
        if (lane == 1) *reinterpret_cast<f32x4 *>(z + orow * 8 + 4) = f32x4{zc[4], zc[5], zc[6], zc[7]};
    }
}

// statistics only: the normalisation itself is applied by the consuming GEMM while it stages its
// A operand (gemm_x3.hip / gemm_h2.hip), so LN(x) is never written to HBM.  One row per wave.
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ x, int64_t rows, int dim, float eps,
                                                        float *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = (dim / 4 + 63) / 64;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + row * dim);
    f32x4 v[LN_MAX_V4];
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (i < nv && (lane + 64 * i) * 4 < dim) v[i] = __builtin_nontemporal_load(xr + lane + 64 * i);
    float mean, rstd;
    row_stats(v, nv, lane, dim, eps, mean, rstd);
    if (lane == 0) *reinterpret_cast<float2 *>(stats + 2 * row) = make_float2(mean, rstd);
}

// Per-clip ranges from the row statistics (clip of row m = m / rows_per_clip): atomic max into
// bounds[clip][2] (zeroed by the caller) of
//   [0] |mean| + sqrt(dim) / rstd   >= max_k |x_k|      (|x_k - mean| <= sqrt(dim * var))
//   [1] sqrt(dim * (mean^2 + 1 / rstd^2)) >= ||x||_2
// -- the range information the fp16-split GEMM needs for operands derived from x (gemm_h2.hip), per clip so
// that a clip's results never depend on what else is in the batch.  Reads only the [rows][2] statistics;
// grid (RANGE_SPLIT, n_clips): RANGE_SPLIT atomics per slot (hot-spot atomics from every row-statistics
// workgroup cost ~5 us per 1000 and were measured at 0.2 ms per launch).
constexpr int RANGE_SPLIT = 8;
__global__ __launch_bounds__(256) void range_from_stats_kernel(const float *__restrict__ stats, int64_t rows,
                                                               int64_t rows_per_clip, int dim,
                                                               float *__restrict__ bounds) {
    const int64_t clip = blockIdx.y;
    const int64_t lo = clip * rows_per_clip, hi = lo + rows_per_clip < rows ? lo + rows_per_clip : rows;
    const float sq = sqrtf((float)dim);
    float b0 = 0.0f, b1 = 0.0f;
    for (int64_t m = lo + blockIdx.x * 256 + threadIdx.x; m < hi; m += (int64_t)RANGE_SPLIT * 256) {
        const float2 ms = *reinterpret_cast<const float2 *>(stats + 2 * m);
        const float sd = 1.0f / ms.y;
        const float r0 = fabsf(ms.x) + sq * sd, r1 = sqrtf((float)dim * (ms.x * ms.x + sd * sd));
        b0 = fmaxf(b0, r0 == r0 ? r0 : 0.0f);  // NaN rows leave the range alone
        b1 = fmaxf(b1, r1 == r1 ? r1 : 0.0f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        b0 = fmaxf(b0, __shfl_xor(b0, o));
        b1 = fmaxf(b1, __shfl_xor(b1, o));
    }
    __shared__ float sh[2][4];
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = b0;
        sh[1][threadIdx.x >> 6] = b1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float m = fmaxf(fmaxf(sh[threadIdx.x][0], sh[threadIdx.x][1]), fmaxf(sh[threadIdx.x][2], sh[threadIdx.x][3]));
        // non-negative floats order like their bit patterns
        atomicMax(reinterpret_cast<unsigned *>(bounds) + 2 * clip + threadIdx.x, __float_as_uint(m));
    }
}

// out[0] = max over rows of ||w_row||_2, out[1] = max |w| (atomic max; the caller zeroes out): the static
// range information of a weight matrix / LayerNorm vector (rows = 1) for gemm_h2.hip's bounds
__global__ __launch_bounds__(256) void weight_range_kernel(const float *__restrict__ w, int64_t ld, int rows, int K,
                                                           float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= rows) return;
    float ss = 0.0f, mx = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float v = w[(int64_t)n * ld + k];
        ss += v * v;
        mx = fmaxf(mx, fabsf(v));
    }
    ss = wave_allsum(ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) {
        atomicMax(reinterpret_cast<unsigned *>(out), __float_as_uint(sqrtf(ss)));
        atomicMax(reinterpret_cast<unsigned *>(out) + 1, __float_as_uint(mx));
    }
}

// one wave per patch: gathers the (c, pt, p1, p2) features (p2 contiguous in memory) and
// layer-normalises them.  feature f = ((c*pt + j)*p + p1)*p + p2.
__global__ __launch_bounds__(256) void patchify_ln_kernel(const float *__restrict__ video, int B, int C, int F, int H,
                                                          int W, int f0, int t, int pt, int p,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float *__restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int h = H / p, w = W / p;
    const int64_t rows = (int64_t)B * t * h * w;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int dim = C * pt * p * p;
    const int nv = (dim / 4 + 63) / 64;
    int64_t rr = row;
    const int pw = (int)(rr % w); rr /= w;
    const int ph = (int)(rr % h); rr /= h;
    const int tt = (int)(rr % t); rr /= t;
    const int b = (int)rr;
    f32x4 v[LN_MAX_V4];
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i) {
        const int f = (lane + 64 * i) * 4;
        if (i < nv && f < dim) {
            int q = f;
            const int p2 = q % p; q /= p;
            const int p1 = q % p; q /= p;
            const int j = q % pt; q /= pt;
            const int c = q;
            const int frame = f0 + tt * pt + j;
            const int64_t off = (((int64_t)(b * C + c) * F + frame) * H + (ph * p + p1)) * W + pw * p + p2;
            v[i] = *reinterpret_cast<const f32x4 *>(video + off);
        }
    }
    f32x4 *yr = reinterpret_cast<f32x4 *>(out + row * ldo);
    // zero the K padding [dim, ldo) the GEMM reads (ldo - dim < 32)
    if (dim + lane * 4 < ldo) yr[dim / 4 + lane] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (!gamma) {  // plain im2col rows (the 'cnn' patch-embed: Conv3d with kernel == stride)
#pragma unroll
        for (int i = 0; i < LN_MAX_V4; ++i)
            if (i < nv && (lane + 64 * i) * 4 < dim) yr[lane + 64 * i] = v[i];
        return;
    }
    float mean, rstd;
    row_stats(v, nv, lane, dim, eps, mean, rstd);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gamma);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(beta);
#pragma unroll
    for (int i = 0; i < LN_MAX_V4; ++i)
        if (i < nv && (lane + 64 * i) * 4 < dim) {
            const f32x4 g = g4[lane + 64 * i], bb = b4[lane + 64 * i];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            yr[lane + 64 * i] = o;
        }
}

__global__ __launch_bounds__(256) void unpatchify_kernel(const float *__restrict__ tok, int B, int C, int F, int H,
                                                         int W, int f0, int t, int pt, int p,
                                                         float *__restrict__ video) {
    const int h = H / p, w = W / p;
    const int dim4 = C * pt * p * p / 4;
    const int64_t total = (int64_t)B * t * h * w * dim4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int f = (int)(idx % dim4) * 4;
    int64_t rr = idx / dim4;
    const int pw = (int)(rr % w); rr /= w;
    const int ph = (int)(rr % h); rr /= h;
    const int tt = (int)(rr % t); rr /= t;
    const int b = (int)rr;
    int q = f;
    const int p2 = q % p; q /= p;
    const int p1 = q % p; q /= p;
    const int j = q % pt; q /= pt;
    const int c = q;
    const int frame = f0 + tt * pt + j;
    const int64_t off = (((int64_t)(b * C + c) * F + frame) * H + (ph * p + p1)) * W + pw * p + p2;
    *reinterpret_cast<f32x4 *>(video + off) = reinterpret_cast<const f32x4 *>(tok)[idx];
}

// y[b, c, a, :] = x[b, a, c, :]
__global__ __launch_bounds__(256) void transpose_tokens_kernel(const f32x4 *__restrict__ x, f32x4 *__restrict__ y,
                                                               int64_t B, int64_t A, int64_t Cn, int d4) {
    const int64_t total = B * A * Cn * d4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % d4);
        int64_t rr = idx / d4;
        const int64_t a = rr % A; rr /= A;
        const int64_t c = rr % Cn;
        const int64_t b = rr / Cn;
        y[idx] = x[((b * A + a) * Cn + c) * d4 + d];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stats + pack: the row statistics of x (mean, rstd: what row_stats_kernel computes) AND x itself as fp16 hi|lo planes
// with one power-of-two scale per row -- the raw-activation operand of the q|k|v plane GEMM (gemm_pl.h; reference
// attention.py:404-412: K and V are projected from the un-normalised x, Q from LayerNorm(x), which the GEMM epilogue
// recovers from the same planes with the statistics).  Replaces the row-statistics pass in front of the attention GEMMs.
// Workgroup = 64 rows (one row block of planes), wave = 16 rows held in registers (H float4 per lane and row, lanes along
// k: coalesced loads); the chunks go through LDS so that every global store instruction writes 4 runs of 256 bytes
// (16 consecutive rows x 16 B of one (k block, plane, k group)).
constexpr int SP_CSTRIDE = 16 * 16 + 16;  // LDS stride of a 16-row chunk column (16 B pad: bank spread)

// win_gw > 0: the OUTPUT rows (planes, a_scale, stats) are in window-major order -- row = (frame, window, position inside the
// ws x ws window), the order reference attention.py:170-188 window_partition produces -- while x is read in token order
// (frame, y, x): the operand of the window-attention q|k|v GEMM, whose packing epilogues then see 64 consecutive rows per
// window and need no row map of their own.
// MODE 2: the planes hold LayerNorm(x) = (x - mean) rstd gamma (+ beta) scaled by the power of two of the static bound ln_bound
// (a_scale is not written): the operand of a plane GEMM that follows a standalone LayerNorm (to_pixels after norm_out).
// The mode is a template parameter: the plain pass (MODE 0) is the hottest HBM-bound kernel of the step and keeps its
// branch-free load batch.
struct SpArgs {
    const float *x;
    int64_t rows;
    float eps;
    int center;
    unsigned char *planes;
    float *a_scale, *stats;
    int win_gh, win_gw, win_ws;
    int t_nseq;  // SP_TEMPORAL: sequences of 5 tokens; output row -> (tile of 64 sequences, half, time step, sequence)
    const float *ln_g, *ln_b;
    float ln_bound;
};
enum { SP_PLAIN = 0, SP_WINDOWS = 1, SP_LAYERNORM = 2, SP_TEMPORAL = 3 };

// RW = rows per wave: 16 (a workgroup = one 64-row plane block), or 4 for small calls (r06: a workgroup = 16 rows, four times the
// workgroups -- one image's 1024 rows are 64 workgroups instead of 16, each a quarter as long; every row's arithmetic is the same)
template <int H, int MODE, int RW = 16>
__global__ __launch_bounds__(256, 2) void stats_pack_kernel(SpArgs a) {
    constexpr int CST = RW * 16 + 16;  // LDS stride of an RW-row chunk column (16 B pad: bank spread)
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    constexpr int K = 256 * H;
    const float *__restrict__ x = a.x;
    const int64_t rows = a.rows;
    const float eps = a.eps;
    const int center = a.center;
    unsigned char *__restrict__ planes = a.planes;
    float *__restrict__ a_scale = a.a_scale;
    float *__restrict__ stats = a.stats;
    const int win_gh = a.win_gh, win_gw = a.win_gw, win_ws = a.win_ws;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * (4 * RW) + wave * RW;
    unsigned char *lds = sp_lds + wave * (2 * 32 * CST);  // [plane][32 chunks][RW rows][16 B] (padded)
    f32x4 v[RW][H];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int64_t row = row0 + r;
        int64_t src = row < rows ? row : rows - 1;
        bool pad = row >= rows;
        if constexpr (MODE == SP_TEMPORAL) {
            // output row = [tile][half][t][32 sequences] (the five time steps of a sequence land in the five row blocks of one
            // lane pair of the temporal GEMM's wave tile, gemm_pl.h PL_TSCORE / PL_TPV); input row = sequence * 5 + t
            const int64_t tile = src / 320;
            const int rem = (int)(src - tile * 320), half = rem / 160, t = (rem % 160) >> 5, j = rem & 31;
            const int64_t seq = tile * 64 + half * 32 + j;
            pad = pad || seq >= a.t_nseq;
            src = (seq < a.t_nseq ? seq : 0) * 5 + t;
        }
        if constexpr (MODE == SP_WINDOWS) {
            const int S = win_gh * win_gw, w2 = win_ws * win_ws, nwx = win_gw / win_ws;
            const int64_t f = src / S;
            const int rr = (int)(src - f * S), win = rr / w2, pos = rr - win * w2;
            const int y = (win / nwx) * win_ws + pos / win_ws, xx = (win % nwx) * win_ws + pos % win_ws;
            src = f * S + y * win_gw + xx;
        }
        const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + src * K);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            v[r][h] = __builtin_nontemporal_load(xr + lane + 64 * h);
            if (pad) v[r][h] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // pad rows of the plane block are zero
        }
    }
    f32x4 lg[H], lb[H];
    if constexpr (MODE == SP_LAYERNORM) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            lg[h] = reinterpret_cast<const f32x4 *>(a.ln_g)[lane + 64 * h];
            lb[h] = a.ln_b ? reinterpret_cast<const f32x4 *>(a.ln_b)[lane + 64 * h] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    float scl[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float s = 0.0f, mx = 0.0f;
#pragma unroll
        for (int h = 0; h < H; ++h) s += (v[r][h][0] + v[r][h][1]) + (v[r][h][2] + v[r][h][3]);
        s = wave_allsum(s);
        const float mean = s / (float)K;
        float q = 0.0f;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float a = v[r][h][0] - mean, b = v[r][h][1] - mean, c = v[r][h][2] - mean, d = v[r][h][3] - mean;
            q += (a * a + b * b) + (c * c + d * d);
            if (center) v[r][h] = f32x4{a, b, c, d};  // x - mean, rounded once like LayerNorm's own first step
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[r][h][0]), fabsf(v[r][h][1])), fmaxf(fabsf(v[r][h][2]), fabsf(v[r][h][3]))));
        }
        q = wave_allsum(q);
        const int64_t row = row0 + r;
        if constexpr (MODE == SP_LAYERNORM) {  // LayerNorm(x) with the arithmetic of layernorm_kernel, one static scale for every row
            const float rstd = 1.0f / sqrtf(q / (float)K + eps);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (center ? v[r][h][e] : v[r][h][e] - mean) * rstd * lg[h][e];
                if (a.ln_b) o += lb[h];
                v[r][h] = row < rows ? o : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            scl[r] = h2_scale_of_bound(a.ln_bound);
            continue;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float sc = h2_scale_of_bound(mx);
        scl[r] = sc;
        if (lane == 0 && row < rows) {
            if (stats) *reinterpret_cast<float2 *>(stats + 2 * row) = make_float2(mean, 1.0f / sqrtf(q / (float)K + eps));
            a_scale[row] = 1.0f / sc;
        }
    }
    const int kblocks = K >> 5;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        // this half's 32 chunks of 8 k: lane pair (2 j, 2 j + 1) holds chunk j
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const f32x4 t = v[r][h] * scl[r];  // exact (power of two)
            const f16x4 hh = __builtin_convertvector(t, f16x4);
            const f16x4 ll = __builtin_convertvector(t - __builtin_convertvector(hh, f32x4), f16x4);
            unsigned char *dst = lds + (lane >> 1) * CST + r * 16 + (lane & 1) * 8;
            *reinterpret_cast<u32x2 *>(dst) = __builtin_bit_cast(u32x2, hh);
            *reinterpret_cast<u32x2 *>(dst + 32 * CST) = __builtin_bit_cast(u32x2, ll);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes (no other wave touches this region)
        __builtin_amdgcn_wave_barrier();
        // stores: lane -> (chunk column cc = lane / RW, row r = lane % RW); 64 / RW chunk columns per instruction
        const int cc = lane / RW, r = lane % RW;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c0 = 0; c0 < 32; c0 += 64 / RW) {
                const int c = c0 + cc;             // chunk inside this half
                const int cg = h * 32 + c;         // chunk of the row: k block cg / 4, k group cg % 4
                const u32x4 w = *reinterpret_cast<const u32x4 *>(lds + (p * 32 + c) * CST + r * 16);
                *reinterpret_cast<u32x4 *>(planes + pl_chunk_offset(row0 + r, cg >> 2, cg & 3, kblocks) + p * 4096) = w;
            }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned *__restrict__ p, unsigned value, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = value;
}

int device_fill_u32(void *ptr, unsigned value, int64_t n_words, hipStream_t stream) {
    if (n_words <= 0) return OMNITOK_OK;
    int64_t blocks = (n_words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<unsigned *>(ptr), value, n_words);
    OT_LAUNCH_CHECK("fill_u32");
    return OMNITOK_OK;
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_layernorm(const float *x, const float *gamma, const float *beta, float *y, int64_t rows,
                                 int dim, float eps, int64_t rows_per_group, int64_t group_stride,
                                 int64_t group_offset, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && gamma && y, "layernorm: null pointer");
    OT_CHECK_ARG(dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAX_V4, "layernorm: dim=%d unsupported", dim);
    OT_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(gamma) && (!beta || aligned16(beta)),
                 "layernorm: pointers must be 16-byte aligned");
    if (rows == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, y,
                       rows, dim, eps, rows_per_group, group_stride, group_offset, 0, 0);
    OT_LAUNCH_CHECK("layernorm");
    return OMNITOK_OK;
}

extern "C" int omnitok_layernorm_transposed(const float *x, const float *gamma, const float *beta, float *y, int64_t n,
                                            int a, int c, int dim, float eps, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && gamma && y && x != y, "layernorm_transposed: null pointer / in-place");
    OT_CHECK_ARG(dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAX_V4 && a > 0 && c > 0, "layernorm_transposed: bad shape");
    OT_CHECK_ARG(aligned16(x) && aligned16(y) && aligned16(gamma) && (!beta || aligned16(beta)),
                 "layernorm_transposed: pointers must be 16-byte aligned");
    const int64_t rows = n * a * c;
    if (rows == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, gamma, beta, y,
                       rows, dim, eps, (int64_t)0, (int64_t)0, (int64_t)0, a, c);
    OT_LAUNCH_CHECK("layernorm_transposed");
    return OMNITOK_OK;
}

extern "C" int omnitok_layernorm_prevq(const float *x, const float *gamma, const float *beta, const float *w, const float *b,
                                       float *z, int64_t n, int a, int c, int dim, float eps, int transpose, int l2,
                                       omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && gamma && w && b && z, "layernorm_prevq: null pointer");
    OT_CHECK_ARG((dim == 256 || dim == 512 || dim == 1024) && a > 0 && c > 0, "layernorm_prevq: dim %d (256, 512 or 1024)", dim);
    OT_CHECK_ARG(aligned16(x) && aligned16(z) && aligned16(gamma) && aligned16(w) && (!beta || aligned16(beta)),
                 "layernorm_prevq: pointers must be 16-byte aligned");
    const int64_t rows = n * a * c;
    if (rows == 0) return OMNITOK_OK;
    const dim3 grid((unsigned)((rows + 4 * PV_ROWS - 1) / (4 * PV_ROWS)));
    const int tr_a = transpose ? a : 0;
    if (dim == 256)
        hipLaunchKernelGGL(layernorm_prevq_kernel<1>, grid, dim3(256), 0, stream, x, gamma, beta, w, b, z, rows, eps, tr_a, c, l2);
    else if (dim == 512)
        hipLaunchKernelGGL(layernorm_prevq_kernel<2>, grid, dim3(256), 0, stream, x, gamma, beta, w, b, z, rows, eps, tr_a, c, l2);
    else
        hipLaunchKernelGGL(layernorm_prevq_kernel<4>, grid, dim3(256), 0, stream, x, gamma, beta, w, b, z, rows, eps, tr_a, c, l2);
    OT_LAUNCH_CHECK("layernorm_prevq");
    return OMNITOK_OK;
}

extern "C" int omnitok_row_stats(const float *x, int64_t rows, int dim, float eps, float *stats, float *bounds,
                                 int64_t rows_per_clip, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && stats, "row_stats: null pointer");
    OT_CHECK_ARG(dim > 0 && dim % 4 == 0 && dim <= 256 * LN_MAX_V4, "row_stats: dim=%d unsupported", dim);
    OT_CHECK_ARG(aligned16(x) && (reinterpret_cast<uintptr_t>(stats) & 7) == 0, "row_stats: alignment");
    if (rows == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, rows, dim, eps,
                       stats);
    if (bounds) {
        const int64_t rpc = rows_per_clip > 0 ? rows_per_clip : rows;
        const int64_t n_clips = (rows + rpc - 1) / rpc;
        OT_CHECK_ARG(n_clips <= 65535, "row_stats: %lld clips (max 65535)", (long long)n_clips);
        hipLaunchKernelGGL(range_from_stats_kernel, dim3(RANGE_SPLIT, (unsigned)n_clips), dim3(256), 0, stream, stats,
                           rows, rpc, dim, bounds);
    }
    OT_LAUNCH_CHECK("row_stats");
    return OMNITOK_OK;
}

static int stats_pack_impl(const float *x, int64_t rows, int dim, float eps, int center, void *planes, int64_t m_pad,
                           float *a_scale, float *stats, float *bounds, int64_t rows_per_clip, int win_gh, int win_gw, int win_ws,
                           omnitok_stream_t stream_);

extern "C" int omnitok_stats_pack(const float *x, int64_t rows, int dim, float eps, int center, void *planes, int64_t m_pad,
                                  float *a_scale, float *stats, float *bounds, int64_t rows_per_clip,
                                  omnitok_stream_t stream_) {
    return stats_pack_impl(x, rows, dim, eps, center, planes, m_pad, a_scale, stats, bounds, rows_per_clip, 0, 0, 0, stream_);
}

extern "C" int omnitok_stats_pack_windows(const float *x, int64_t rows, int dim, float eps, int center, void *planes,
                                          int64_t m_pad, float *a_scale, float *stats, int gh, int gw, int ws,
                                          omnitok_stream_t stream_) {
    OT_CHECK_ARG(gh > 0 && gw > 0 && ws > 0 && gh % ws == 0 && gw % ws == 0 && rows % ((int64_t)gh * gw) == 0,
                 "stats_pack_windows: %lld rows of a %d x %d grid, window %d", (long long)rows, gh, gw, ws);
    return stats_pack_impl(x, rows, dim, eps, center, planes, m_pad, a_scale, stats, nullptr, 0, gh, gw, ws, stream_);
}

namespace omnitok {
int g_sp_small_blocks = 0;  // "sp_small_blocks": 0 = stats_pack in its 4-rows-per-wave form at every size | n: only below n 64-row blocks
}  // namespace omnitok

template <int MODE>
static int stats_pack_launch(SpArgs a, int dim, int64_t m_pad, hipStream_t stream) {
    OT_CHECK_ARG(dim > 0 && dim % 256 == 0 && dim <= 1024, "stats_pack: dim=%d (multiples of 256 up to 1024)", dim);
    if (m_pad == 0) return OMNITOK_OK;
    // 4 rows per wave (16-row workgroups) at every size since r06: written for small calls (one image: 19 -> 9 us per launch), it also
    // wins where 64-row workgroups make ragged rounds -- 8 clips 0.84 -> 0.54 ms per step, C5 1.30 -> 1.00, C3 2.15 -> 1.89
    // (profiles/r06_small_calls.txt).  "sp_small_blocks" n > 0: only below n 64-row blocks (1 = the 16-rows-per-wave form everywhere).
    const bool small = g_sp_small_blocks > 0 ? m_pad / 64 < g_sp_small_blocks : true;
    const int lds = small ? 4 * 2 * 32 * (4 * 16 + 16) : 4 * 2 * 32 * SP_CSTRIDE;
    const dim3 grid((unsigned)(small ? m_pad / 16 : m_pad / 64));
#define OT_SP(Hh)                                                                                                          \
    do {                                                                                                                   \
        if (small) {                                                                                                       \
            hipLaunchKernelGGL((stats_pack_kernel<Hh, MODE, 4>), grid, dim3(256), lds, stream, a);                         \
        } else {                                                                                                           \
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(stats_pack_kernel<Hh, MODE>), lds)) return rc; \
            hipLaunchKernelGGL((stats_pack_kernel<Hh, MODE>), grid, dim3(256), lds, stream, a);                            \
        }                                                                                                                  \
    } while (0)
    switch (dim / 256) {
        case 1: OT_SP(1); break;
        case 2: OT_SP(2); break;
        case 3: OT_SP(3); break;
        default: OT_SP(4); break;
    }
#undef OT_SP
    OT_LAUNCH_CHECK("stats_pack");
    return OMNITOK_OK;
}

static int stats_pack_impl(const float *x, int64_t rows, int dim, float eps, int center, void *planes, int64_t m_pad,
                           float *a_scale, float *stats, float *bounds, int64_t rows_per_clip, int win_gh, int win_gw, int win_ws,
                           omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && planes && a_scale, "stats_pack: null pointer");
    OT_CHECK_ARG(m_pad >= rows && m_pad % 64 == 0 && aligned16(x) && aligned16(planes) &&
                     (!stats || (reinterpret_cast<uintptr_t>(stats) & 7) == 0) && (!bounds || stats),
                 "stats_pack: m_pad %% 64, alignment, bounds need stats");
    SpArgs a{};
    a.x = x; a.rows = rows; a.eps = eps; a.center = center; a.planes = static_cast<unsigned char *>(planes);
    a.a_scale = a_scale; a.stats = stats; a.win_gh = win_gh; a.win_gw = win_gw; a.win_ws = win_ws;
    if (int rc = (win_gw > 0 ? stats_pack_launch<SP_WINDOWS>(a, dim, m_pad, stream) : stats_pack_launch<SP_PLAIN>(a, dim, m_pad, stream)))
        return rc;
    if (bounds && rows > 0) {
        const int64_t rpc = rows_per_clip > 0 ? rows_per_clip : rows;
        const int64_t n_clips = (rows + rpc - 1) / rpc;
        OT_CHECK_ARG(n_clips <= 65535, "stats_pack: %lld clips (max 65535)", (long long)n_clips);
        hipLaunchKernelGGL(range_from_stats_kernel, dim3(RANGE_SPLIT, (unsigned)n_clips), dim3(256), 0, stream, stats, rows, rpc,
                           dim, bounds);
        OT_LAUNCH_CHECK("range_from_stats");
    }
    return OMNITOK_OK;
}

// The same pass for the fused temporal stage (gemm_pl.h PL_TSCORE / PL_TPV): x [nseq * 5, dim] in token order (sequence, time
// step); planes, a_scale and stats in the permuted row order [tile of 64 sequences][half][time step][32 sequences],
// ceil(nseq / 64) * 320 rows (sequences beyond nseq are zero rows).  bounds per clip of seqs_per_clip sequences (% 64 == 0).
extern "C" int omnitok_stats_pack_temporal(const float *x, int64_t nseq, int dim, float eps, void *planes, float *a_scale,
                                           float *stats, float *bounds, int64_t seqs_per_clip, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && planes && a_scale && stats && nseq >= 0 && nseq < (1ll << 31) / 5, "stats_pack_temporal: bad arguments");
    OT_CHECK_ARG(aligned16(x) && aligned16(planes) && (reinterpret_cast<uintptr_t>(stats) & 7) == 0, "stats_pack_temporal: alignment");
    OT_CHECK_ARG(!bounds || (seqs_per_clip > 0 && seqs_per_clip % 64 == 0 && nseq % seqs_per_clip == 0),
                 "stats_pack_temporal: per-clip ranges need whole clips of a multiple of 64 sequences");
    if (nseq == 0) return OMNITOK_OK;
    const int64_t rows = (nseq + 63) / 64 * 320;
    SpArgs a{};
    a.x = x; a.rows = rows; a.eps = eps; a.center = 1; a.planes = static_cast<unsigned char *>(planes);
    a.a_scale = a_scale; a.stats = stats; a.t_nseq = (int)nseq;
    if (int rc = stats_pack_launch<SP_TEMPORAL>(a, dim, rows, stream)) return rc;
    if (bounds) {
        const int64_t rpc = seqs_per_clip * 5, n_clips = nseq / seqs_per_clip;
        OT_CHECK_ARG(n_clips <= 65535, "stats_pack_temporal: %lld clips (max 65535)", (long long)n_clips);
        hipLaunchKernelGGL(range_from_stats_kernel, dim3(RANGE_SPLIT, (unsigned)n_clips), dim3(256), 0, stream, stats, rows, rpc, dim,
                           bounds);
        OT_LAUNCH_CHECK("range_from_stats");
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_layernorm_planes(const float *x, int64_t rows, int dim, float eps, const float *gamma, const float *beta,
                                        float bound, void *planes, int64_t m_pad, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && gamma && planes && bound > 0.0f, "layernorm_planes: null pointer / bound");
    OT_CHECK_ARG(m_pad >= rows && m_pad % 64 == 0 && aligned16(x) && aligned16(planes) && aligned16(gamma) && (!beta || aligned16(beta)),
                 "layernorm_planes: m_pad %% 64, alignment");
    SpArgs a{};
    a.x = x; a.rows = rows; a.eps = eps; a.planes = static_cast<unsigned char *>(planes);
    a.ln_g = gamma; a.ln_b = beta; a.ln_bound = bound;
    return stats_pack_launch<SP_LAYERNORM>(a, dim, m_pad, stream);
}

extern "C" int omnitok_weight_range(const float *w, int64_t ldw, int rows, int K, float *out2,
                                    omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(w && out2 && rows > 0 && K > 0, "weight_range: bad arguments");
    hipLaunchKernelGGL(weight_range_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, w, ldw, rows, K, out2);
    OT_LAUNCH_CHECK("weight_range");
    return OMNITOK_OK;
}

extern "C" int omnitok_patchify_ln(const float *video, int B, int C, int F, int H, int W, int f0, int t, int pt,
                                   int p, const float *gamma, const float *beta, float eps, float *out,
                                   int64_t ldo, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(video && out && ((gamma != nullptr) == (beta != nullptr)), "patchify_ln: null pointer");
    OT_CHECK_ARG(p % 4 == 0 && H % p == 0 && W % p == 0, "patchify_ln: patch size %d / image %dx%d unsupported", p,
                 H, W);
    OT_CHECK_ARG(f0 >= 0 && f0 + t * pt <= F, "patchify_ln: frame range out of bounds");
    const int dim = C * pt * p * p;
    OT_CHECK_ARG(dim <= 256 * LN_MAX_V4, "patchify_ln: feature dim %d too large", dim);
    if (ldo == 0) ldo = dim;
    OT_CHECK_ARG(ldo >= dim && ldo % 4 == 0 && ldo - dim < 256, "patchify_ln: bad output row stride %lld", (long long)ldo);
    const int64_t rows = (int64_t)B * t * (H / p) * (W / p);
    if (rows == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(patchify_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, video, B, C, F, H,
                       W, f0, t, pt, p, gamma, beta, eps, out, ldo);
    OT_LAUNCH_CHECK("patchify_ln");
    return OMNITOK_OK;
}

extern "C" int omnitok_unpatchify(const float *tok, int B, int C, int F, int H, int W, int f0, int t, int pt, int p,
                                  float *video, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(tok && video, "unpatchify: null pointer");
    OT_CHECK_ARG(p % 4 == 0 && H % p == 0 && W % p == 0, "unpatchify: patch size %d unsupported", p);
    OT_CHECK_ARG(f0 >= 0 && f0 + t * pt <= F, "unpatchify: frame range out of bounds");
    const int64_t total = (int64_t)B * t * (H / p) * (W / p) * (C * pt * p * p / 4);
    if (total == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, tok, B, C, F,
                       H, W, f0, t, pt, p, video);
    OT_LAUNCH_CHECK("unpatchify");
    return OMNITOK_OK;
}

extern "C" int omnitok_transpose_tokens(const float *x, float *y, int64_t B, int64_t A, int64_t C, int D,
                                        omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && y && x != y, "transpose_tokens: bad pointers");
    OT_CHECK_ARG(D % 4 == 0, "transpose_tokens: D %% 4 != 0");
    const int64_t total = B * A * C * (D / 4);
    if (total == 0) return OMNITOK_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(transpose_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4 *>(x), reinterpret_cast<f32x4 *>(y), B, A, C, D / 4);
    OT_LAUNCH_CHECK("transpose_tokens");
    return OMNITOK_OK;
}
