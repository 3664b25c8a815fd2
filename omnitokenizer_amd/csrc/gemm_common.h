// Shared pieces of the GEMM kernels (fp32-MFMA kernels in gemm.hip, split-operand kernels in gemm_x3.hip /
// gemm_h2.hip): parameters, tile order, LDS-only barrier, fused epilogue.
#pragma once
#include "common.h"

namespace omnitok {

constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;  // LDT: padded LDS row length (floats)
constexpr int TILE_FLOATS = BM * LDT;
constexpr int GEMM_LDS_BYTES = 2 /*buffers*/ * 2 /*A,B*/ * TILE_FLOATS * 4;

struct GemmParams {
    const float *a;
    const float *w;
    const float *bias;
    const float *residual;
    float *c;
    int64_t lda, ldw, ldr, ldc;
    int64_t M;
    int N, K;
    int nbn;  // number of tile columns
    int nbm;  // number of tile rows
    int ntiles;
    long long *trace;  // debug: per-wave s_memtime stamps of workgroup 0 (null = off)
    int64_t a_rpg, a_stride, a_off;
    int gn;  // tile columns per column group of the tile order ("gemm_gn" option, default 8)
};

// exact-erf GELU (reference attention.py:155-156, F.gelu default).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. fp32-roundoff class) in 14 branch-free VALU instructions instead of the
// library erff's ~34 + divergent branches: the GEGLU epilogue evaluates it 32 times per lane per
// tile, and on gfx950 VALU work between fp32 MFMAs costs matrix-pipe time.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
    return copysignf(fmaf(-p, e, 1.0f), x);
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

// Fused epilogue on the accumulator registers (bias, leaky-relu, residual add, GEGLU).
// NEDGE = false (N % 64 == 0, every production shape): buffer loads/stores through wave-uniform
// descriptors of this wave's 64x64 output tile -- per element one SGPR offset and one shared 32-bit
// lane offset, so no 64-bit VGPR address per element, no VALU address arithmetic, and all residual
// loads of a column block in flight together.  Rows beyond M are dropped by the descriptor's
// num_records bound (loads return 0), so no per-element guards are needed: with guards hipcc emits
// branch / load / wait / store per element, i.e. 64 serialised memory round trips per tile.
// NEDGE = true: per-element guards (only the N = heads = 8 GEMM of the position-bias MLP).
template <int FLAGS, bool NEDGE>
__device__ __forceinline__ void gemm_epilogue(const GemmParams &p, f32x16 (&acc)[2][2], int64_t bm, int bn, int wm,
                                              int wn, int r32, int hi) {
    const int64_t row0 = bm * BM + wm * 64;
    auto uniform_rsrc = [](const float *ptr, int bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi32 << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    int64_t vr64 = p.M - row0;  // valid rows of this wave tile
    const int vr = vr64 > 64 ? 64 : (vr64 < 0 ? 0 : (int)vr64);
    if constexpr (FLAGS & OMNITOK_GEMM_GEGLU) {
        const int ocolb = (bn * 2 + wn) * 32;
        const int ldc = (int)p.ldc;
        const auto c_rs = uniform_rsrc(p.c + row0 * p.ldc + ocolb, vr > 0 ? ((vr - 1) * ldc + 32) * 4 : 0);
        const int c_voff = (4 * hi * ldc + r32) * 4;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(
                    __builtin_bit_cast(unsigned, gelu_erf(acc[mb][1][r]) * acc[mb][0][r]), c_rs, c_voff,
                    (mb * 32 + mfma32_row(r, 0)) * ldc * 4, 0);
    } else if constexpr (!NEDGE) {
        const int colb = bn * BN + wn * 64;
        if (colb >= p.N) return;  // N % 64 == 0: a wave tile is entirely inside or entirely outside
        const int ldc = (int)p.ldc, ldr = (int)p.ldr;
        const auto c_rs = uniform_rsrc(p.c + row0 * p.ldc + colb, vr > 0 ? ((vr - 1) * ldc + 64) * 4 : 0);
        const int c_voff = (4 * hi * ldc + r32) * 4;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float res[2][16];  // 32 residual loads in flight per column block
            if constexpr (FLAGS & OMNITOK_GEMM_RESIDUAL) {
                const auto r_rs = uniform_rsrc(p.residual + row0 * p.ldr + colb, vr > 0 ? ((vr - 1) * ldr + 64) * 4 : 0);
                const int r_voff = (4 * hi * ldr + r32) * 4;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[mb][r] = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(
                                       r_rs, r_voff, ((mb * 32 + mfma32_row(r, 0)) * ldr + nb * 32) * 4, 0));
            }
            float bv = 0.0f;
            if constexpr (FLAGS & OMNITOK_GEMM_BIAS) bv = p.bias[colb + nb * 32 + r32];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mb][nb][r];
                    if constexpr (FLAGS & OMNITOK_GEMM_BIAS) v += bv;
                    if constexpr (FLAGS & OMNITOK_GEMM_LEAKY) v = v > 0.0f ? v : 0.1f * v;
                    if constexpr (FLAGS & OMNITOK_GEMM_RESIDUAL) v += res[mb][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, c_voff,
                                                          ((mb * 32 + mfma32_row(r, 0)) * ldc + nb * 32) * 4, 0);
                }
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = bn * BN + wn * 64 + nb * 32 + r32;
            const bool colok = col < p.N;
            float bv = 0.0f;
            if constexpr (FLAGS & OMNITOK_GEMM_BIAS) bv = colok ? p.bias[col] : 0.0f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = row0 + mb * 32 + mfma32_row(r, hi);
                    if (row < p.M && colok) {
                        float v = acc[mb][nb][r];
                        if constexpr (FLAGS & OMNITOK_GEMM_BIAS) v += bv;
                        if constexpr (FLAGS & OMNITOK_GEMM_LEAKY) v = v > 0.0f ? v : 0.1f * v;
                        if constexpr (FLAGS & OMNITOK_GEMM_RESIDUAL) v += p.residual[row * p.ldr + col];
                        p.c[row * p.ldc + col] = v;
                    }
                }
        }
    }
}

// XCD-aware bijective remap of a linear tile id (workgroup b runs on XCD b % 8): each XCD sweeps a
// contiguous range of the tile order below, so neighbouring tiles share that XCD's L2
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int xcd = id & 7, q = n >> 3, rem = n & 7;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
}

// Tile order: column groups of GN = 8 tiles, row-major inside a group.  The ~64 tiles an XCD has
// in flight then touch 8 A panels + 8 W tiles (4 MiB at K = 512 = the XCD's L2) instead of 3 A
// panels + ALL W tiles (5.8 MiB for the FF-in weight, which thrashed: 4.5 GB of memory-side reads
// per launch against 0.34 GB of operands).
__device__ __forceinline__ void tile_coords(int lid, int nbm, int nbn, int GN, int64_t &bm, int &bn) {
    const int full = nbn / GN;            // number of full column groups
    const int gsz = GN * nbm;             // tiles per full group
    int cg = lid / gsz, rem = lid - cg * gsz, width = GN;
    if (cg >= full) {                     // the last, narrower group
        cg = full;
        rem = lid - full * gsz;
        width = nbn - full * GN;
    }
    bm = rem / width;
    bn = cg * GN + rem % width;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would
// force the global loads issued in an L phase to land before the phase can end (they are not
// needed until the next L phase, two matrix phases later).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only (vmcnt / expcnt fields at their maximum)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace omnitok
