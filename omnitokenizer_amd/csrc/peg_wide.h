// PEG ring kernel with a 64-channel slab: 256-byte pieces per position instead of peg3d_lds_kernel's 128 -- a plain copy of the
// token tensor in 128-byte pieces runs 15 % below one in >= 256-byte pieces (profiles/r04_hbm_shape_probe.txt).  Included by
// peg.hip and by the stand-alone probe tools/probes/peg_wide_probe.hip (bit equality + time against the in-tree kernel).
// A workgroup of 256 threads owns a 4 x 16 tile of 64 channels (thread = channel quad x column, 4 output rows) and walks the time
// axis.  Input-stationary: a plane is read from LDS once, when it arrives, and feeds the three output planes it belongs to (tap
// dt = 2 of the oldest, 1 of the middle, 0 of the youngest), so two ring slots (the plane in use + the one being stored) replace
// the ring kernel's three and the LDS reads fall to a third.  Every output still receives its planes in the order dt = 0, 1, 2 and
// its taps in the order (r, dw, dh): bit-identical to peg3d_lds_kernel.  Needs D % 64 == 0, W % 16 == 0, H % 4 == 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace omnitok {
namespace pegw {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H_T = 4, W_T = 16, C4 = 16;                     // tile rows, cols, float4 channels (64 channels)
constexpr int HW = W_T + 2, HH = H_T + 2;
constexpr int PLANE_V4 = HH * HW * C4;                        // 1728 float4 = 27648 B
constexpr int LOADS = (PLANE_V4 + 255) / 256;                 // 7 per thread (the last one covers 192 threads)
constexpr int LDS_BYTES = 2 * PLANE_V4 * 16;                  // 55296: two workgroups per CU

__device__ __attribute__((aligned(16))) float peg_wide_zero[4] = {0.f, 0.f, 0.f, 0.f};

__global__ __launch_bounds__(256, 2) void peg3d_wide_kernel(const float *__restrict__ x, const float *__restrict__ w27,
                                                             const float *__restrict__ bias, float *__restrict__ y,
                                                             int B, int T, int H, int W, int D, int tpad) {
    extern __shared__ __attribute__((aligned(16))) float pegw_smem[];
    f32x4 *ring = reinterpret_cast<f32x4 *>(pegw_smem);
    const int tid = threadIdx.x;
    const int c4 = tid & 15, wcol = tid >> 4;  // 16 channel quads x 16 columns
    const int wtiles = W / W_T, htiles = H / H_T;
    // XCD-aware order (workgroup n runs on XCD n % 8): the tiles of one (clip, slab) are consecutive workgroups of one XCD
    const int per = wtiles * htiles, ngroups = gridDim.x / per;
    const int n = blockIdx.x;
    int grp, tile;
    {
        const int xcd = n & 7, j = n >> 3;
        const int full = (ngroups / 8) * 8;
        const int g = xcd + 8 * (j / per);
        if (g < full) {
            grp = g;
            tile = j % per;
        } else {
            const int rest = n - full * per;
            grp = full + rest / per;
            tile = rest % per;
        }
    }
    const int w0 = (tile % wtiles) * W_T, h0 = (tile / wtiles) * H_T;
    const int nslab = D >> 6;
    const int slab = grp % nslab, b = grp / nslab;
    const int d4n = D >> 2;
    const int ch4 = slab * C4 + c4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);

    f32x4 wt[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wt[k] = reinterpret_cast<const f32x4 *>(w27)[k * d4n + ch4];
    const f32x4 bv = reinterpret_cast<const f32x4 *>(bias)[ch4];

    // this thread's elements of a halo plane tile, as float4 offsets inside a plane (-1: outside the grid -> zero); same for every plane
    int poff[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
        const int i = tid + 256 * k;
        const int pos = i >> 4, q = i & 15;
        const int r = pos / HW, c = pos - r * HW;
        const int hh = h0 - 1 + r, ww = w0 - 1 + c;
        poff[k] = (i < PLANE_V4 && hh >= 0 && hh < H && ww >= 0 && ww < W) ? (hh * W + ww) * d4n + slab * C4 + q : -1;
    }
    const int64_t plane_v4 = (int64_t)H * W * d4n;
    const f32x4 *xb = x4 + (int64_t)b * T * plane_v4;
    const f32x4 *zero = reinterpret_cast<const f32x4 *>(peg_wide_zero);
    f32x4 stage[LOADS];
    auto gload = [&](int tau) {
        const f32x4 *xp = xb + (int64_t)tau * plane_v4;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) stage[k] = *(poff[k] >= 0 ? xp + poff[k] : zero);  // unconditional loads (no exec branches)
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const int i = tid + 256 * k;
            if (i < PLANE_V4) ring[slot * PLANE_V4 + i] = stage[k];
        }
    };

    f32x4 o0[H_T], o1[H_T], o2[H_T];  // outputs t0 = tau + tpad - 2 (completes this step), t0 + 1, t0 + 2
#pragma unroll
    for (int r = 0; r < H_T; ++r) o0[r] = o1[r] = o2[r] = bv;
    const int n_iter = T + 2 - tpad;
    gload(0);
    for (int tau = 0; tau < n_iter; ++tau) {
        if (tau < T) lstore(tau & 1);  // the slot held plane tau - 2: last read in step tau - 1 (its centre taps, non-causal case)
        __syncthreads();
        if (tau + 1 < T) gload(tau + 1);  // in flight while this step computes
        const int t0 = tau + tpad - 2;
        if (tau < T && w0 + wcol < W) {
            const f32x4 *pl = ring + (tau & 1) * PLANE_V4;
#pragma unroll
            for (int r = 0; r < HH; ++r) {
                f32x4 v[3];
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) v[dw] = pl[(r * HW + wcol + dw) * C4 + c4];
                // the three targets keep their own (r, dw, dh) order; they are independent sums
                if (t0 >= 0) {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                        for (int dh = 0; dh < 3; ++dh)
                            if (r - dh >= 0 && r - dh < H_T) o0[r - dh] += v[dw] * wt[(2 * 3 + dh) * 3 + dw];
                }
                if (t0 + 1 >= 0 && t0 + 1 < T) {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                        for (int dh = 0; dh < 3; ++dh)
                            if (r - dh >= 0 && r - dh < H_T) o1[r - dh] += v[dw] * wt[(1 * 3 + dh) * 3 + dw];
                }
                if (t0 + 2 < T) {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                        for (int dh = 0; dh < 3; ++dh)
                            if (r - dh >= 0 && r - dh < H_T) o2[r - dh] += v[dw] * wt[(0 * 3 + dh) * 3 + dw];
                }
                if (r & 1) __builtin_amdgcn_sched_barrier(0);  // two rows of LDS reads in flight, not all six (registers)
            }
        }
        if (t0 >= 0 && t0 < T) {
            // + residual: the centre taps of plane t0 (this step's plane when causal, the previous one otherwise: still in its slot)
            const f32x4 *pc = ring + (t0 & 1) * PLANE_V4;
            f32x4 *yo = reinterpret_cast<f32x4 *>(y) + ((((int64_t)b * T + t0) * H + h0) * W + w0 + wcol) * d4n + ch4;
#pragma unroll
            for (int r = 0; r < H_T; ++r) {
                const f32x4 xc = pc[((r + 1) * HW + wcol + 1) * C4 + c4];
                yo[(int64_t)r * W * d4n] = o0[r] + xc;
            }
        }
#pragma unroll
        for (int r = 0; r < H_T; ++r) {
            o0[r] = o1[r];
            o1[r] = o2[r];
            o2[r] = bv;
        }
        __syncthreads();  // every wave is done with both slots' reads of this step before the next plane is stored
    }
}

// One plane per clip (T == 1: images, reference attention.py:298-338 on a [B, 1, H, W, D] view): the walk above degenerates to
// load -> compute -> store with two workgroups per CU and nothing overlapping (0.40 of HBM peak at C2).  Here only the
// plane's own 9 taps (time tap dt = 2 causal / 1 non-causal) and ONE output set are live: ~100 registers and one ring slot, so
// four workgroups per CU overlap each other's phases.  Same tile, same loader, and every output receives its taps in the order
// (dh, dw) the walk gives them: bit-identical to peg3d_wide_kernel / peg3d_lds_kernel.
constexpr int LDS_BYTES_2D = PLANE_V4 * 16;                   // 27648: up to five workgroups per CU

__global__ __launch_bounds__(256, 4) void peg2d_wide_kernel(const float *__restrict__ x, const float *__restrict__ w27,
                                                             const float *__restrict__ bias, float *__restrict__ y,
                                                             int B, int H, int W, int D, int dt) {
    extern __shared__ __attribute__((aligned(16))) float pegw_smem[];
    f32x4 *pl = reinterpret_cast<f32x4 *>(pegw_smem);
    const int tid = threadIdx.x;
    const int c4 = tid & 15, wcol = tid >> 4;
    const int wtiles = W / W_T, htiles = H / H_T;
    const int per = wtiles * htiles, ngroups = gridDim.x / per;
    const int n = blockIdx.x;
    int grp, tile;
    {
        const int xcd = n & 7, j = n >> 3;
        const int full = (ngroups / 8) * 8;
        const int g = xcd + 8 * (j / per);
        if (g < full) {
            grp = g;
            tile = j % per;
        } else {
            const int rest = n - full * per;
            grp = full + rest / per;
            tile = rest % per;
        }
    }
    const int w0 = (tile % wtiles) * W_T, h0 = (tile / wtiles) * H_T;
    const int nslab = D >> 6;
    const int slab = grp % nslab, b = grp / nslab;
    const int d4n = D >> 2;
    const int ch4 = slab * C4 + c4;
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x) + (int64_t)b * H * W * d4n;
    const f32x4 *zero = reinterpret_cast<const f32x4 *>(peg_wide_zero);
    f32x4 stage[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
        const int i = tid + 256 * k;
        const int pos = i >> 4, q = i & 15;
        const int r = pos / HW, c = pos - r * HW;
        const int hh = h0 - 1 + r, ww = w0 - 1 + c;
        const bool in = i < PLANE_V4 && hh >= 0 && hh < H && ww >= 0 && ww < W;
        stage[k] = *(in ? xb + (hh * W + ww) * d4n + slab * C4 + q : zero);
    }
    f32x4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = reinterpret_cast<const f32x4 *>(w27)[(dt * 9 + k) * d4n + ch4];
    const f32x4 bv = reinterpret_cast<const f32x4 *>(bias)[ch4];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
        const int i = tid + 256 * k;
        if (i < PLANE_V4) pl[i] = stage[k];
    }
    __syncthreads();
    f32x4 o[H_T];
#pragma unroll
    for (int r = 0; r < H_T; ++r) o[r] = bv;
#pragma unroll
    for (int r = 0; r < HH; ++r) {
        f32x4 v[3];
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) v[dw] = pl[(r * HW + wcol + dw) * C4 + c4];
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
                if (r - dh >= 0 && r - dh < H_T) o[r - dh] += v[dw] * wt[dh * 3 + dw];
    }
    f32x4 *yo = reinterpret_cast<f32x4 *>(y) + (((int64_t)b * H + h0) * W + w0 + wcol) * d4n + ch4;
#pragma unroll
    for (int r = 0; r < H_T; ++r) {
        const f32x4 xc = pl[((r + 1) * HW + wcol + 1) * C4 + c4];
        yo[(int64_t)r * W * d4n] = o[r] + xc;
    }
}
}  // namespace pegw
}  // namespace omnitok
