// Vector quantiser: pre_vq projection + l2norm, nearest-codebook search (bit-exact with the
// reference's fp32 arithmetic), de-quantisation + post_vq projection.
//
// vq_argmin (reference modules/codebook.py:82-86; oracle/vq_argmin.c is the arithmetic spec):
//   d[n,c] = (xx[n] - dot[n,c]) + ee[c],   ids[n] = first argmin_c
// codebook_dim = 8 makes the sweep fp32-FMA-bound (~1400 flop/byte), not HBM-bound.
// The dot products run on the matrix pipe: v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf
// chain (C = 0, k = 0..7 over four MFMA steps), i.e. exactly what MKL's sgemm produces for
// (2x) @ E^T on the reference CPU path, while the VALU does the (xx - dot) + ee / compare /
// select epilogue concurrently.  D[code][row]: each lane owns one input row (column of D) and
// sees its codes in increasing index order, so a strict '<' keeps the first minimum; the two
// half-waves are merged with a (distance, index) lexicographic compare.
//   * codebook pre-packed in A-fragment order (one coalesced 16-byte load per lane per 32 codes,
//     L2-resident: 256 KiB for 8192 codes), ee[] staged in LDS and read as broadcast float4;
//   * each wave keeps two 32-row B fragments, so one A load feeds 8 MFMAs.
#include "common.h"
#include "h2_common.h"

namespace omnitok {

constexpr int VQ_ROWS_PER_WAVE = 64;
constexpr int VQ_ROWS_PER_BLOCK = 256;
int g_vq_split = 0;  // 0 = automatic
// "vq_variant" (VQ_CODEBOOK mode; A/B record in profiles/r03_vq_variants.txt):
//   0  dot on the matrix pipe, (xx - dot) + ee on the VALU, codebook fragments from L2            (rounds 1-2)
//   1  fifth MFMA step: the chain runs on -2z (exactly -dot), then k = 8: 1 * xx, k = 9: ee * 1 -- the accumulator
//      ends in fl(fl(xx - dot) + ee), the reference's distance bit for bit, and the epilogue is the min tree alone
//      (default: 0.268 vs 0.284 ms at C3)
//   2  like 0 with the split's codebook fragments staged in LDS once per workgroup (1 KiB per 32 codes): 0.32 ms, the
//      L2-resident fragments (one coalesced 16-byte load per lane and tile, prefetched two tiles ahead) were never the limit
int g_vq_variant = 1;
// "vq_screen" 1 (default): omnitok_encode's nearest-code search is the screened form (vq_screen_kernel below: same ids,
// a fifth of the matrix work) | 0: the exact sweep for every code.  "vq_screen_split": 0 = automatic, else forced.
int g_vq_screen = 1;
int g_vq_screen_split = 0;

__global__ void vq_prepare_kernel(const float *__restrict__ E, int n_codes, float *__restrict__ packed,
                                  float *__restrict__ ee) {
    // packed float4 index = tile*64 + lane, lane = hi*32 + r32 -> {E[c][hi], E[c][2+hi], E[c][4+hi], E[c][6+hi]}
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_codes * 2) return;
    const int tile = idx >> 6, lane = idx & 63, r32 = lane & 31, hi = lane >> 5;
    const int c = tile * 32 + r32;
    f32x4 a;
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = E[c * 8 + 2 * s + hi];
    reinterpret_cast<f32x4 *>(packed)[idx] = a;
    if (hi == 0) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __fadd_rn(acc, __fmul_rn(E[c * 8 + k], E[c * 8 + k]));
        ee[c] = acc;
    }
}

// SPLIT: the code range is divided over gridDim.y workgroups per row block (load balance: 64 rows x
// all codes is ~150 us of work per wave, so whole-range units quantise badly over the chip's wave
// slots -- 2560 units on 2048 slots at C3 -- and leave small batches on a handful of CUs).  Each split
// merges its (distance, index) into ids[row] with a 64-bit atomicMin on (order-preserving distance
// key << 32 | index): the lexicographic minimum is exactly "first minimum" (reference codebook.py:86),
// whatever the arrival order.  ids must be pre-set to all ones; vq_finalize_kernel strips the key.
__device__ __forceinline__ unsigned order_key(float d) {
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// MODE: which distance the sweep minimises (all: first minimum, k-ordered FMA dot from 0):
//   VQ_CODEBOOK  modules/codebook.py:82-86                 d = (xx - 2 dot) + ee
//   VQ_COS       external CosineSimCodebook (vector_quantize_pytorch.py:646-650): first argmax of the dot.
//                xx = ee = 0: d = (0 - 2 dot) + 0 = -2 dot exactly (doubling is exact, order preserving)
//   VQ_EUCLID    external EuclideanCodebook (:29-33, :463): first argmax of -cdist = first argmin of
//                sqrt(clamp((xx + ee) + (-2 dot), 0)); the correctly rounded sqrt is part of the ordering
//                (distinct squared distances can round to the same root -> tie -> lowest index)
enum { VQ_CODEBOOK = 0, VQ_COS = 1, VQ_EUCLID = 2 };

template <bool SPLIT, int MODE = VQ_CODEBOOK, int VAR = 0>
__global__ __launch_bounds__(256, VAR == 2 ? 1 : 2) void vq_argmin_kernel(const float *__restrict__ z,
                                                           const float *__restrict__ packed,
                                                           const float *__restrict__ ee_g, int64_t n, int n_codes,
                                                           int64_t *__restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) float ee_s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int ntiles_all = n_codes >> 5;
    const int tiles_per = (ntiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t0 = SPLIT ? (int)blockIdx.y * tiles_per : 0;
    const int t1 = SPLIT ? (t0 + tiles_per < ntiles_all ? t0 + tiles_per : ntiles_all) : ntiles_all;
    if (t0 >= t1) return;
    const int c0 = t0 * 32;  // first code of this split; ee_s holds ee[c0 .. t1*32)
    constexpr bool FIFTH = VAR == 1 && MODE == VQ_CODEBOOK;
    constexpr bool CB_LDS = VAR == 2;
    // LDS: ee[c0 .. t1 * 32) and, for VAR 2, the codebook fragments of the same codes behind them
    f32x4 *cb_s = reinterpret_cast<f32x4 *>(ee_s + (MODE != VQ_COS ? ((t1 - t0) * 32 + 3) / 4 * 4 : 0));
    if (MODE != VQ_COS) {
        for (int i = tid * 4; i < (t1 - t0) * 32; i += 256 * 4)
            *reinterpret_cast<f32x4 *>(ee_s + i) = *reinterpret_cast<const f32x4 *>(ee_g + c0 + i);
    }
    if (CB_LDS) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(packed) + (int64_t)t0 * 64;
        for (int i = tid; i < (t1 - t0) * 64; i += 256) cb_s[i] = src[i];
    }
    if (MODE != VQ_COS || CB_LDS) __syncthreads();

    const int64_t row_base = (int64_t)blockIdx.x * VQ_ROWS_PER_BLOCK + wave * VQ_ROWS_PER_WAVE;
    float xb[2][4], xx[2], best[2];
    f32x4 zlo[2], zhi[2];  // 2 z of the lane's rows (all 8 channels): the end-of-sweep rescan needs the full row
    int btile[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int64_t row = row_base + g * 32 + r32;
        if (row > n - 1) row = n - 1;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(z + row * 8);
        const f32x4 hi4 = *reinterpret_cast<const f32x4 *>(z + row * 8 + 4);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(lo[k], lo[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(hi4[k], hi4[k]));
        xx[g] = MODE == VQ_COS ? 0.0f : acc;
        zlo[g] = lo * 2.0f;  // exact
        zhi[g] = hi4 * 2.0f;
        // k = 2s + hi for MFMA step s.  FIFTH: the chain runs on -2z -- every step is fl(a * (-b) + c) = -fl(a * b - c),
        // round-to-nearest is sign-symmetric, so the accumulator is exactly -dot
        const float sg = FIFTH ? -1.0f : 1.0f;
        xb[g][0] = sg * (hi ? zlo[g][1] : zlo[g][0]);
        xb[g][1] = sg * (hi ? zlo[g][3] : zlo[g][2]);
        xb[g][2] = sg * (hi ? zhi[g][1] : zhi[g][0]);
        xb[g][3] = sg * (hi ? zhi[g][3] : zhi[g][2]);
        best[g] = INFINITY;
        btile[g] = t0;
    }
    // FIFTH, MFMA step 4 (k = 8 + hi): k = 8 is 1 * xx -> fl(xx - dot), k = 9 is ee[code] * 1 -> fl(fl(xx - dot) + ee)
    float xb5[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) xb5[g] = hi ? 1.0f : xx[g];
    auto a5_of = [&](int t) { return hi ? ee_s[t * 32 + r32 - c0] : 1.0f; };
    auto dist = [&](float xxg, float dot, float ee) {
        if (MODE == VQ_EUCLID)  // + 0.0f: sqrt(-0.0) = -0.0 must not order below +0.0 in the split key
            return __fadd_rn(__fsqrt_rn(fmaxf(__fsub_rn(__fadd_rn(xxg, ee), dot), 0.0f)), 0.0f);
        return __fadd_rn(__fsub_rn(xxg, dot), ee);
    };

    // Sweep: per 32-code tile only the lane's minimum distance (v_min3 tree, ~2.5 VALU per element instead of
    // the 5 of a running (value, index) pair) and the tile that holds it; strict '<' keeps the FIRST tile of a tie.
    // Software pipeline: the 8 MFMAs of tile t + 1 are issued with the epilogue of tile t (two accumulator sets),
    // so a wave's VALU work sits beside its own matrix instructions instead of after them.
    const f32x4 *pk = reinterpret_cast<const f32x4 *>(packed) + lane;
    auto ld_a = [&](int t) -> f32x4 { return CB_LDS ? cb_s[(t - t0) * 64 + lane] : pk[(int64_t)t * 64]; };
    auto dots = [&](const f32x4 &a, float a5, f32x16 (&acc)[2]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], xb[g][s], acc[g], 0, 0, 0);
            if (FIFTH) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a5, xb5[g], acc[g], 0, 0, 0);
        }
    };
    // one quarter (q) of a tile's epilogue: the lane's 4 codes cbase + 8 q + (0..3) of both row groups
    auto epi_quarter = [&](int t, int q, const f32x16 (&acc)[2], float (&tmin)[2]) {
        const int cbase = t * 32 + 4 * hi;
        const f32x4 e4 = MODE == VQ_COS ? f32x4{0.0f, 0.0f, 0.0f, 0.0f}
                             : *reinterpret_cast<const f32x4 *>(ee_s + cbase - c0 + 8 * q);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float d0, d1, d2, d3;
            if constexpr (FIFTH) {  // the accumulators ARE the distances
                d0 = acc[g][q * 4 + 0]; d1 = acc[g][q * 4 + 1]; d2 = acc[g][q * 4 + 2]; d3 = acc[g][q * 4 + 3];
            } else if constexpr (MODE == VQ_EUCLID) {
                d0 = dist(xx[g], acc[g][q * 4 + 0], e4[0]); d1 = dist(xx[g], acc[g][q * 4 + 1], e4[1]);
                d2 = dist(xx[g], acc[g][q * 4 + 2], e4[2]); d3 = dist(xx[g], acc[g][q * 4 + 3], e4[3]);
            } else {
                // (xx - dot) + ee on register pairs: v_pk_add_f32 rounds each half like the scalar add
                // (IEEE, no contraction), so the distances are bit-identical at half the issue slots
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 x2 = {xx[g], xx[g]};
                const f32x2 a01 = {acc[g][q * 4 + 0], acc[g][q * 4 + 1]}, a23 = {acc[g][q * 4 + 2], acc[g][q * 4 + 3]};
                const f32x2 e01 = {e4[0], e4[1]}, e23 = {e4[2], e4[3]};
                const f32x2 r01 = (x2 - a01) + e01, r23 = (x2 - a23) + e23;
                d0 = r01[0]; d1 = r01[1]; d2 = r23[0]; d3 = r23[1];
            }
            tmin[g] = fminf(fminf(tmin[g], d0), d1);
            tmin[g] = fminf(fminf(tmin[g], d2), d3);
        }
    };
    auto commit = [&](int t, const float (&tmin)[2]) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (tmin[g] < best[g]) {
                best[g] = tmin[g];
                btile[g] = t;
            }
    };
    auto epilogue = [&](int t, const f32x16 (&acc)[2]) {
        float tmin[2] = {INFINITY, INFINITY};
#pragma unroll
        for (int q = 0; q < 4; ++q) epi_quarter(t, q, acc, tmin);
        commit(t, tmin);
    };
    // MFMA step s of the NEXT tile (both row groups) followed by quarter s of the CURRENT tile's epilogue; the
    // scheduling barriers pin that issue order (the wave issues in order: eight MFMAs back to back would hold the
    // VALU work behind the matrix pipe, and the compiler otherwise groups them)
    auto fused = [&](const f32x4 &a, float a5, f32x16 (&nxt)[2], int t, const f32x16 (&cur)[2]) {
        float tmin[2] = {INFINITY, INFINITY};
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) nxt[g][r] = 0.0f;
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
            nxt[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sq], xb[0][sq], nxt[0], 0, 0, 0);
            nxt[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sq], xb[1][sq], nxt[1], 0, 0, 0);
            if (FIFTH && sq == 3) {
                nxt[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a5, xb5[0], nxt[0], 0, 0, 0);
                nxt[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a5, xb5[1], nxt[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            epi_quarter(t, sq, cur, tmin);
            __builtin_amdgcn_sched_barrier(0);
        }
        commit(t, tmin);
    };
    f32x16 accA[2], accB[2];
    f32x4 a_cur = ld_a(t0);
    dots(a_cur, FIFTH ? a5_of(t0) : 0.0f, accA);
    int t = t0;
    for (; t + 2 < t1; t += 2) {   // tiles t, t + 1 complete here; t + 2 is in flight at the end
        const f32x4 a1 = ld_a(t + 1);
        const f32x4 a2 = ld_a(t + 2);
        const float e1 = FIFTH ? a5_of(t + 1) : 0.0f, e2 = FIFTH ? a5_of(t + 2) : 0.0f;
        fused(a1, e1, accB, t, accA);
        fused(a2, e2, accA, t + 1, accB);
    }
    if (t + 1 < t1) {              // two tiles left: t (in flight) and t + 1
        const f32x4 a1 = ld_a(t + 1);
        fused(a1, FIFTH ? a5_of(t + 1) : 0.0f, accB, t, accA);
        epilogue(t + 1, accB);
    } else {
        epilogue(t, accA);
    }
    // Resolve the index: redo the lane's 16 codes of its winning tile with the same arithmetic -- the fp32 MFMA is
    // bitwise the k-ordered fmaf chain from 0 (oracle/vq_argmin.c), so the scalar chain reproduces the distances --
    // in increasing code order with a strict '<' (first minimum inside the tile).
    int bidx[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int t = btile[g];
        const float *pt = packed + (int64_t)t * 256;  // float index of code r (half hh) step s: (hh*32 + r)*4 + s
        float bd = INFINITY;
        int bi = t * 32 + 4 * hi;
        const float zz[8] = {zlo[g][0], zlo[g][1], zlo[g][2], zlo[g][3], zhi[g][0], zhi[g][1], zhi[g][2], zhi[g][3]};
#pragma unroll 4
        for (int qe = 0; qe < 16; ++qe) {
            const int r = 8 * (qe >> 2) + (qe & 3) + 4 * hi;  // row of accumulator register qe inside the tile
            const f32x4 ev = *reinterpret_cast<const f32x4 *>(pt + r * 4);        // k = 0, 2, 4, 6
            const f32x4 od = *reinterpret_cast<const f32x4 *>(pt + (32 + r) * 4);  // k = 1, 3, 5, 7
            float dot = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dot = __fmaf_rn(ev[s], zz[2 * s], dot);
                dot = __fmaf_rn(od[s], zz[2 * s + 1], dot);
            }
            const float ee = MODE == VQ_COS ? 0.0f : ee_s[t * 32 + r - c0];
            const float d = dist(xx[g], dot, ee);
            if (d < bd) {
                bd = d;
                bi = t * 32 + r;
            }
        }
        best[g] = bd;
        bidx[g] = bi;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float od = swap32(best[g]);
        const int oi = __shfl_xor(bidx[g], 32);
        if (od < best[g] || (od == best[g] && oi < bidx[g])) {
            best[g] = od;
            bidx[g] = oi;
        }
        const int64_t row = row_base + g * 32 + r32;
        if (hi == 0 && row < n) {
            if (SPLIT)
                atomicMin(reinterpret_cast<unsigned long long *>(ids + row),
                          ((unsigned long long)order_key(best[g]) << 32) | (unsigned)bidx[g]);
            else
                ids[row] = (int64_t)bidx[g];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SCREENED nearest-code search ("vq_variant" 3, VQ_CODEBOOK distance; reference modules/codebook.py:82-86): the same ids as
// vq_argmin_kernel, bit for bit, at a fraction of its matrix work.
//
// The exact search spends five fp32-input MFMAs (320 matrix cycles) per 32 codes x 32 rows because the distance must be
// the reference's own fp32 value fl(fl(xx - dot) + ee) with dot a k-ordered fmaf chain -- that is the tie-breaking
// contract.  But almost every code is nowhere near the minimum, and for THOSE any cheap distance with a rigorous error
// bound is enough to rule them out.  One fp16 MFMA (v_mfma_f32_32x32x16_f16, 32 matrix cycles) evaluates
//     s'[c] = ee_h[c] + ee_l[c] - sum_k fp16(E[c][k]) * fp16(2 z[k])          (K = 16: 8 products, ee as two halves x 1)
// for a whole tile: the A fragment of lane (r32, half) is half 0: fp16(E[c][0..7]), half 1: (ee_h, ee_l, 0 ...); the B
// fragment half 0: fp16(-2 z[0..7]), half 1: (1, 1, 0 ...).  With u = 2^-11 (fp16 unit roundoff), eta = 2^-25 (half the
// subnormal spacing), |E_c| <= M_E and ee_c <= M_ee for every code (vq_screen_prepare), n_z >= |z|_2:
//     |s'[c] - (ee_c - 2 z . E_c)| <= eps := 4.002 u M_E n_z                       products of two rounded factors (Cauchy-Schwarz)
//                                          + 1.001 eta sqrt(8) (M_E + 2 n_z) + 2^-47    subnormal inputs
//                                          + 2^-22 M_ee + 2^-25                         ee = ee_h + ee_l + rest
//                                          + 2^-18 (M_ee + 2.01 M_E n_z + 1)            ten exact fp16 x fp16 products summed in fp32
//                                                                                       (64 units in the last place of the largest
//                                                                                       partial sum: any adder tree qualifies)
// and the exact path's own distance differs from the real number xx + (ee_c - 2 z . E_c) by at most
//     delta := 2^-24 * 1.001 * (21 M_E n_z + 2 xx + M_ee)       (gamma_8 of the fmaf chain + the two roundings of the epilogue).
// Let c* be the reference's answer (first minimum of the fp32 distance d) and c_m the minimiser of s'.  Then
//     s'[c*] <= d[c*] - xx + delta + eps <= d[c_m] - xx + delta + eps <= s'[c_m] + 2 (delta + eps),
// i.e. c* -- and every code tied with it in fp32 -- lies within tau = 2 (eps + delta) of the screen's minimum.
// Pass 1 sweeps all codes for min s' (per row); pass 2 sweeps again and RECORDS the tiles holding a code within tau of it
// (almost always exactly one); the recorded tiles are then re-evaluated with the exact fmaf chain of the kernel above, in
// increasing code order with a strict '<'.  Rows whose inputs leave fp16's range or are not finite, and rows with more than
// SCR_CAP candidate tiles (a codebook full of duplicates), re-evaluate every tile: slow, never wrong.
constexpr int SCR_CAP = 4;

__global__ void vq_screen_prepare_kernel(const float *__restrict__ E, const float *__restrict__ ee, int n_codes,
                                         u32x4 *__restrict__ frag, unsigned *__restrict__ consts) {
    // consts[0] = max |E_c|_2 (rounded up), [1] = max ee_c, [2] = max |E_ck| as float bits (non-negative: uint order)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_codes * 2) return;
    const int tile = idx >> 6, lane = idx & 63, r32 = lane & 31, hi = lane >> 5;
    const int c = tile * 32 + r32;
    f16x8 h;
    if (hi == 0) {
        float nrm2 = 0.0f, amax = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = E[c * 8 + k];
            h[k] = (_Float16)v;
            nrm2 = fmaf(v, v, nrm2);
            amax = fmaxf(amax, fabsf(v));
        }
        float nr = sqrtf(nrm2) * 1.000001f;
        if (!(nr >= 0.0f)) nr = INFINITY;      // NaN in the codebook: every row takes the exact path
        if (!(amax >= 0.0f)) amax = INFINITY;
        atomicMax(consts + 0, __float_as_uint(nr));
        atomicMax(consts + 2, __float_as_uint(amax));
    } else {
        float e = ee[c];
        if (!(e >= 0.0f)) e = INFINITY;
        atomicMax(consts + 1, __float_as_uint(e));
        const _Float16 eh = (_Float16)ee[c];
        const _Float16 el = (_Float16)(ee[c] - (float)eh);
        h = f16x8{eh, el, 0, 0, 0, 0, 0, 0};
    }
    frag[idx] = __builtin_bit_cast(u32x4, h);
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void vq_screen_kernel(const float *__restrict__ z, const float *__restrict__ packed,
                                                           const float *__restrict__ ee_g, const u32x4 *__restrict__ frag,
                                                           const float *__restrict__ consts, int64_t n, int n_codes,
                                                           int64_t *__restrict__ ids) {
    extern __shared__ __attribute__((aligned(16))) float ee_s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int ntiles_all = n_codes >> 5;
    const int tiles_per = (ntiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t0 = SPLIT ? (int)blockIdx.y * tiles_per : 0;
    const int t1 = SPLIT ? (t0 + tiles_per < ntiles_all ? t0 + tiles_per : ntiles_all) : ntiles_all;
    if (t0 >= t1) return;
    const int c0 = t0 * 32;
    for (int i = tid * 4; i < (t1 - t0) * 32; i += 256 * 4)
        *reinterpret_cast<f32x4 *>(ee_s + i) = *reinterpret_cast<const f32x4 *>(ee_g + c0 + i);
    __syncthreads();

    const float nE = consts[0], nEE = consts[1], nAbs = consts[2];
    const bool cb_ok = nAbs <= 60000.0f && nEE <= 60000.0f && nE <= 60000.0f;  // false for inf / NaN too
    const int64_t row_base = (int64_t)blockIdx.x * VQ_ROWS_PER_BLOCK + wave * VQ_ROWS_PER_WAVE;
    float xx[2], thr[2];
    f32x4 zlo[2], zhi[2];
    f16x8 bf[2];
    bool all_tiles[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int64_t row = row_base + g * 32 + r32;
        if (row > n - 1) row = n - 1;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(z + row * 8);
        const f32x4 hi4 = *reinterpret_cast<const f32x4 *>(z + row * 8 + 4);
        float acc = 0.0f, amax = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(lo[k], lo[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(hi4[k], hi4[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k) amax = fmaxf(amax, fmaxf(fabsf(lo[k]), fabsf(hi4[k])));
        xx[g] = acc;
        zlo[g] = lo * 2.0f;  // exact
        zhi[g] = hi4 * 2.0f;
        // fp16 range of 2 z, finiteness (fmaxf drops NaN, so xx is tested too)
        all_tiles[g] = !(cb_ok && amax <= 30000.0f && acc <= 1e9f);
        if (hi == 0)
            bf[g] = f16x8{(_Float16)(-zlo[g][0]), (_Float16)(-zlo[g][1]), (_Float16)(-zlo[g][2]), (_Float16)(-zlo[g][3]),
                          (_Float16)(-zhi[g][0]), (_Float16)(-zhi[g][1]), (_Float16)(-zhi[g][2]), (_Float16)(-zhi[g][3])};
        else
            bf[g] = f16x8{(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
        // tau = 2 (eps + delta), see the header; every constant rounded up
        const float nz = sqrtf(acc) * 1.00001f + 1e-30f;
        const float men = nE * nz;
        const float eps = 4.002f * 4.8828125e-4f * men + 1.001f * 2.98023224e-8f * 2.8285f * (nE + 2.0f * nz) + 7.2e-15f +
                          2.38418579e-7f * nEE + 2.98023224e-8f + 3.81469727e-6f * (nEE + 2.01f * men + 1.0f);
        const float delta = 5.96046448e-8f * 1.001f * (21.0f * men + 2.0f * acc + nEE);
        thr[g] = 2.0f * (eps + delta) * 1.01f;  // becomes min s' + tau after pass 1
    }
    const u32x4 *pf = frag + lane;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto tile_min = [&](const f32x16 &a) {
        float m = fminf(fminf(a[0], a[1]), a[2]);
        m = fminf(fminf(m, a[3]), a[4]);
        m = fminf(fminf(m, a[5]), a[6]);
        m = fminf(fminf(m, a[7]), a[8]);
        m = fminf(fminf(m, a[9]), a[10]);
        m = fminf(fminf(m, a[11]), a[12]);
        m = fminf(fminf(m, a[13]), a[14]);
        return fminf(m, a[15]);
    };
    auto screen = [&](const u32x4 &a, int g) {
        return tile_min(__builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), bf[g], zero16, 0, 0, 0));
    };
    // ---- pass 1: the screen's minimum per row ----------------------------------------------------------------------
    // Fragments are requested two tiles ahead into four rotating registers (no copies: a copy would wait for the load it
    // moves); clamped indices keep every load unconditional (a load inside a branch makes the compiler's waitcnt pass drain
    // vmcnt(0) at the top of every iteration).
    auto ld = [&](int t) { return pf[(int64_t)(t < t1 ? t : t1 - 1) * 64]; };
    float mn[2] = {INFINITY, INFINITY};
    {
        auto step = [&](const u32x4 &a) {
            mn[0] = fminf(mn[0], screen(a, 0));
            mn[1] = fminf(mn[1], screen(a, 1));
        };
        u32x4 A = ld(t0), B = ld(t0 + 1);
        for (int t = t0; t < t1; t += 4) {
            const u32x4 C = ld(t + 2), D = ld(t + 3);
            step(A);
            if (t + 1 < t1) step(B);
            A = ld(t + 4);
            B = ld(t + 5);
            if (t + 2 < t1) step(C);
            if (t + 3 < t1) step(D);
        }
    }
    int cnt[2] = {0, 0};
    int ct[2][SCR_CAP];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        mn[g] = fminf(mn[g], swap32(mn[g]));  // the other half-wave holds the other 16 codes of every tile
        // |min s'| enters through the rounding of the sum; a NaN / inf minimum (no finite screen value) -> every tile
        thr[g] = mn[g] + (thr[g] + 1.2e-7f * fabsf(mn[g]));
        if (!(fabsf(thr[g]) < INFINITY)) all_tiles[g] = true;
#pragma unroll
        for (int i = 0; i < SCR_CAP; ++i) ct[g][i] = t0;
    }
    // ---- pass 2: record the tiles that hold a code within tau of it ------------------------------------------------
    {
        auto step = [&](const u32x4 &a, int t) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float m = screen(a, g);
                if (m <= thr[g]) {
#pragma unroll
                    for (int i = 0; i < SCR_CAP; ++i)
                        if (cnt[g] == i) ct[g][i] = t;
                    ++cnt[g];
                }
            }
        };
        u32x4 A = ld(t0), B = ld(t0 + 1);
        for (int t = t0; t < t1; t += 4) {
            const u32x4 C = ld(t + 2), D = ld(t + 3);
            step(A, t);
            if (t + 1 < t1) step(B, t + 1);
            A = ld(t + 4);
            B = ld(t + 5);
            if (t + 2 < t1) step(C, t + 2);
            if (t + 3 < t1) step(D, t + 3);
        }
    }
    // ---- exact re-evaluation of the recorded tiles (the arithmetic of vq_argmin_kernel's index resolution) ----------
    float best[2];
    int bidx[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        // a lane whose own 16 codes were never within tau still takes part in the merge below with (inf, first code)
        const bool every = all_tiles[g] || cnt[g] > SCR_CAP;
        // both half-waves must scan the same tiles for the merge to see every candidate: a tile recorded by one half only
        // is simply absent from the other half's list -- its codes there are above the threshold, hence not the answer
        const int nt = every ? t1 - t0 : cnt[g];
        float bd = INFINITY;
        int bi = t0 * 32 + 4 * hi;
        const float zz[8] = {zlo[g][0], zlo[g][1], zlo[g][2], zlo[g][3], zhi[g][0], zhi[g][1], zhi[g][2], zhi[g][3]};
        for (int i = 0; __builtin_amdgcn_ballot_w64(i < nt) != 0; ++i) {
            if (i < nt) {
                int t = t0 + i;
                if (!every) {
                    t = ct[g][0];
#pragma unroll
                    for (int j = 1; j < SCR_CAP; ++j)
                        if (i == j) t = ct[g][j];
                }
                const float *pt = packed + (int64_t)t * 256;
#pragma unroll 4
                for (int qe = 0; qe < 16; ++qe) {
                    const int r = 8 * (qe >> 2) + (qe & 3) + 4 * hi;
                    const f32x4 ev = *reinterpret_cast<const f32x4 *>(pt + r * 4);
                    const f32x4 od = *reinterpret_cast<const f32x4 *>(pt + (32 + r) * 4);
                    float dot = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        dot = __fmaf_rn(ev[s], zz[2 * s], dot);
                        dot = __fmaf_rn(od[s], zz[2 * s + 1], dot);
                    }
                    const float d = __fadd_rn(__fsub_rn(xx[g], dot), ee_s[t * 32 + r - c0]);
                    if (d < bd) {  // tiles and codes in increasing order: strict '<' keeps the first minimum
                        bd = d;
                        bi = t * 32 + r;
                    }
                }
            }
        }
        best[g] = bd;
        bidx[g] = bi;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float od = swap32(best[g]);
        const int oi = __shfl_xor(bidx[g], 32);
        if (od < best[g] || (od == best[g] && oi < bidx[g])) {
            best[g] = od;
            bidx[g] = oi;
        }
        const int64_t row = row_base + g * 32 + r32;
        if (hi == 0 && row < n) {
            if (SPLIT)
                atomicMin(reinterpret_cast<unsigned long long *>(ids + row),
                          ((unsigned long long)order_key(best[g]) << 32) | (unsigned)bidx[g]);
            else
                ids[row] = (int64_t)bidx[g];
        }
    }
}

__global__ __launch_bounds__(256) void vq_finalize_kernel(int64_t *__restrict__ ids, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids[i] &= 0xFFFFFFFFll;
}

// z = l2norm(x W^T + b): one 16-lane DPP row per token row
__global__ __launch_bounds__(256) void pre_vq_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ b, float *__restrict__ z, int64_t n,
                                                     int D, int l2) {
    const int l16 = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (row >= n) return;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
    const float *xr = x + row * D;
    for (int i = l16 * 4; i < D; i += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + i);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c * D + i);
            acc[c] += prevq_dot4(xv[0], xv[1], xv[2], xv[3], wv[0], wv[1], wv[2], wv[3]);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = row16_allsum(acc[c]) + b[c];
    if (l2) prevq_l2norm8(acc);  // F.normalize(p=2, eps=1e-12), omnitokenizer.py:252
    if (l16 == 0) *reinterpret_cast<f32x4 *>(z + row * 8) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    if (l16 == 1) *reinterpret_cast<f32x4 *>(z + row * 8 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// tok[n, 4c..4c+3] = b + sum_k E[ids[n]][k] * W[c][k]
__global__ __launch_bounds__(256) void dequant_post_vq_kernel(const int64_t *__restrict__ ids,
                                                              const float *__restrict__ E, int n_codes,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ b, float *__restrict__ tok,
                                                              int64_t n, int D, int *err_flag) {
    const int d4n = D >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * d4n) return;
    const int64_t row = gid / d4n;
    const int c4 = (int)(gid % d4n);
    int64_t id = ids[row];
    if (id < 0 || id >= n_codes) {
        if (err_flag) atomicOr(err_flag, 1);
        id = 0;
    }
    const f32x4 e0 = *reinterpret_cast<const f32x4 *>(E + id * 8);
    const f32x4 e1 = *reinterpret_cast<const f32x4 *>(E + id * 8 + 4);
    f32x4 o = reinterpret_cast<const f32x4 *>(b)[c4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + (c4 * 4 + j) * 8);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(w + (c4 * 4 + j) * 8 + 4);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(e0[k], w0[k], s);
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(e1[k], w1[k], s);
        o[j] += s;
    }
    reinterpret_cast<f32x4 *>(tok)[gid] = o;
}

// tok[n, :] = table[ids[n], :]: the decode-side form of dequant_post_vq once table = E W^T + b has been built (by
// dequant_post_vq_kernel itself on ids = 0..n_codes-1, so the rows are bit-identical).  Pure row gather from an
// L2 / Infinity-Cache resident table (16 MiB at 8192 x 512): one 16-byte load + store per thread.
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t *__restrict__ ids,
                                                          const f32x4 *__restrict__ table, int n_codes,
                                                          f32x4 *__restrict__ tok, int64_t n, int d4n, int *err_flag,
                                                          int tr_a, int tr_c) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * d4n) return;
    const int64_t row = gid / d4n;
    const int c4 = (int)(gid % d4n);
    int64_t id = ids[row];
    if (id < 0 || id >= n_codes) {
        if (err_flag) atomicOr(err_flag, 1);
        id = 0;
    }
    int64_t orow = row;
    if (tr_a > 0) {  // token transpose fused into the store: row (b, a, c) -> (b, c, a)
        const int64_t c = row % tr_c, ba = row / tr_c;
        orow = ((ba / tr_a) * tr_c + c) * tr_a + ba % tr_a;
    }
    __builtin_nontemporal_store(table[id * d4n + c4], tok + orow * d4n + c4);
}

__global__ void iota_i64_kernel(int64_t *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

__global__ __launch_bounds__(256) void vq_embed_st_kernel(const int64_t *__restrict__ ids,
                                                          const float *__restrict__ z, const float *__restrict__ E,
                                                          int64_t B, int64_t thw, float *__restrict__ emb) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * thw) return;
    const int64_t b = gid / thw, s = gid % thw;
    const int64_t id = ids[gid];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float zv = z[gid * 8 + c];
        emb[(b * 8 + c) * thw + s] = __fadd_rn(__fsub_rn(E[id * 8 + c], zv), zv);  // (e - z) + z
    }
}

// ---- --use_vae: DiagonalGaussianDistribution posterior (reference modules/vae.py:4-17) ---------
// h = x W^T + b with W [16, D] (mean rows 0..7 | logvar rows 8..15, torch.chunk over channels);
// z = mean + exp(0.5 * clamp(logvar, -30, 20)) * noise.  One 16-lane DPP row per token row, lane c
// and lane 8+c end up holding channel c's mean / logvar.  Outputs are channel-FIRST
// [B, 8, thw] like the reference's 'b c t h w' tensors; noise has the same layout (it is
// torch.randn(mean.shape) there).  noise == nullptr -> z = mean (posterior.mode()).
__global__ __launch_bounds__(256) void vae_sample_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ b,
                                                         const float *__restrict__ noise, float *__restrict__ z,
                                                         float *__restrict__ moments, int64_t n, int64_t thw,
                                                         int D) {
    const int l16 = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (row >= n) return;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    const float *xr = x + row * D;
    for (int i = l16 * 4; i < D; i += 64) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + i);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c * D + i);
            acc[c] += (xv[0] * wv[0] + xv[1] * wv[1]) + (xv[2] * wv[2] + xv[3] * wv[3]);
        }
    }
    float mine = 0.0f, other = 0.0f;  // lane l keeps h[l] and h[l ^ 8]
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float v = row16_allsum(acc[c]) + b[c];
        if (c == l16) mine = v;
        if (c == (l16 ^ 8)) other = v;
    }
    const int64_t bi = row / thw, s = row % thw;
    if (moments) moments[(bi * 16 + l16) * thw + s] = mine;
    if (l16 < 8) {
        float out = mine;
        if (noise) {
            const float logvar = fminf(fmaxf(other, -30.0f), 20.0f);
            out = __fadd_rn(mine, __fmul_rn(expf(0.5f * logvar), noise[(bi * 8 + l16) * thw + s]));
        }
        z[(bi * 8 + l16) * thw + s] = out;
    }
}

// tok[n, 4c..4c+3] = b + sum_k z[n][k] * W[c][k] on continuous latents (the --use_vae decode,
// reference omnitokenizer.py:293-317).  z element (row, k): channel-last [n, 8] or channel-first
// [B, 8, thw].
template <bool CHANNEL_FIRST>
__global__ __launch_bounds__(256) void post_vq_latent_kernel(const float *__restrict__ z, int64_t thw,
                                                             const float *__restrict__ w,
                                                             const float *__restrict__ b, float *__restrict__ tok,
                                                             int64_t n, int D) {
    const int d4n = D >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * d4n) return;
    const int64_t row = gid / d4n;
    const int c4 = (int)(gid % d4n);
    float e[8];
    if (CHANNEL_FIRST) {
        const int64_t bi = row / thw, s = row % thw;
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = z[(bi * 8 + k) * thw + s];
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = z[row * 8 + k];
    }
    f32x4 o = reinterpret_cast<const f32x4 *>(b)[c4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + (c4 * 4 + j) * 8);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(w + (c4 * 4 + j) * 8 + 4);
        float sacc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sacc = fmaf(e[k], w0[k], sacc);
#pragma unroll
        for (int k = 0; k < 4; ++k) sacc = fmaf(e[4 + k], w1[k], sacc);
        o[j] += sacc;
    }
    reinterpret_cast<f32x4 *>(tok)[gid] = o;
}

// ---- codebook statistics of Codebook.forward (reference modules/codebook.py:122-140) ----------
// counts[c] = number of tokens mapped to code c (one int atomic per token; ids are spread over
// 8192 codes so contention is negligible)
__global__ __launch_bounds__(256) void vq_hist_kernel(const int64_t *__restrict__ ids, int64_t n, int n_codes,
                                                      int *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        if (id >= 0 && id < n_codes) atomicAdd(counts + id, 1);
    }
}

// one workgroup: batch_usage[c] = counts[c] / n (calculate_batch_codebook_usage_percentage, :54-72),
// perplexity = exp(-sum_c p_c log(p_c + 1e-10)) (:122-123), EMA codebook_usage (:133-136) and
// avg_usage = #(codebook_usage > 1/n_codes) / n_codes (:139-140).  out = {perplexity, avg_usage}.
__global__ __launch_bounds__(1024) void vq_stats_finalize_kernel(const int *__restrict__ counts, int64_t n,
                                                                 int n_codes, float *__restrict__ batch_usage,
                                                                 float *__restrict__ codebook_usage, int first_call,
                                                                 float sigma, float *__restrict__ out) {
    __shared__ float s_ent[1024];
    __shared__ int s_used[1024];
    float ent = 0.0f;
    int used = 0;
    const float thr = 1.0f / (float)n_codes;
    for (int c = threadIdx.x; c < n_codes; c += blockDim.x) {
        const float pc = (float)counts[c] / (float)n;
        batch_usage[c] = pc;
        ent += pc * logf(pc + 1e-10f);
        const float u = first_call ? pc : sigma * codebook_usage[c] + (1.0f - sigma) * pc;
        codebook_usage[c] = u;
        used += u > thr ? 1 : 0;
    }
    s_ent[threadIdx.x] = ent;
    s_used[threadIdx.x] = used;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            s_ent[threadIdx.x] += s_ent[threadIdx.x + st];
            s_used[threadIdx.x] += s_used[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = expf(-s_ent[0]);
        out[1] = (float)s_used[0] / (float)n_codes;
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_vq_stats(const int64_t *ids, int64_t n, int n_codes, int *counts_scratch, float *batch_usage,
                                float *codebook_usage, int first_call, float usage_sigma, float *out2,
                                omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(ids && counts_scratch && batch_usage && codebook_usage && out2 && n > 0 && n_codes > 0,
                 "vq_stats: bad arguments");
    if (int rc = device_fill_u32(counts_scratch, 0u, n_codes, stream)) return rc;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(vq_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ids, n, n_codes, counts_scratch);
    OT_LAUNCH_CHECK("vq_hist");
    hipLaunchKernelGGL(vq_stats_finalize_kernel, dim3(1), dim3(1024), 0, stream, counts_scratch, n, n_codes,
                       batch_usage, codebook_usage, first_call, usage_sigma, out2);
    OT_LAUNCH_CHECK("vq_stats_finalize");
    return OMNITOK_OK;
}

extern "C" int omnitok_pre_vq(const float *x, const float *w, const float *b, float *z, int64_t n, int D, int cdim,
                              int l2, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && w && b && z, "pre_vq: null pointer");
    OT_CHECK_ARG(cdim == 8, "pre_vq: codebook_dim=%d (kernels are built for 8)", cdim);
    OT_CHECK_ARG(D % 64 == 0, "pre_vq: D %% 64 != 0");
    if (n == 0) return OMNITOK_OK;
    const int64_t threads = n * 16;
    hipLaunchKernelGGL(pre_vq_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, w, b, z, n, D,
                       l2);
    OT_LAUNCH_CHECK("pre_vq");
    return OMNITOK_OK;
}

extern "C" int omnitok_vq_prepare(const float *codebook, int n_codes, int cdim, float *packed, float *ee,
                                  omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(codebook && packed && ee, "vq_prepare: null pointer");
    OT_CHECK_ARG(cdim == 8 && n_codes % 32 == 0 && n_codes > 0, "vq_prepare: need cdim == 8, n_codes %% 32 == 0");
    hipLaunchKernelGGL(vq_prepare_kernel, dim3((n_codes * 2 + 255) / 256), dim3(256), 0, stream, codebook, n_codes,
                       packed, ee);
    OT_LAUNCH_CHECK("vq_prepare");
    return OMNITOK_OK;
}

template <int MODE, int VAR>
static int launch_vq_var(const float *z, const float *packed, const float *ee, int64_t n, int n_codes, int64_t *ids,
                         hipStream_t stream) {
    constexpr bool cosine = MODE == VQ_COS;
    const int64_t blocks = (n + VQ_ROWS_PER_BLOCK - 1) / VQ_ROWS_PER_BLOCK;
    // Code-range splits.  Measured at the BASELINE sizes (profiles/r04_vq_split_sweep.txt: C3 640 row blocks x 256 tiles,
    // C2 256 x 256, C5 272 x 512, 8 C5 clips 2176 x 512, one clip 20 x 256, one image 4 x 256): an unsplit launch is the
    // slowest arm at every size but the smallest (one tail round of long workgroups), and past the point where the
    // chip's 512 slots see a few rounds more splits only add prologues and atomics.  The rule minimises
    //     (rounds + 1/2) * (tiles per workgroup + 10),   rounds = row blocks * splits / 512
    // over splits in {1, 2, 4, 8, 16} with at least 16 tiles each -- it picks 4 / 4 / 8 / 2 / 16 / 16 there: the best
    // measured arm at five of the six sizes (one clip: 16 instead of 8, 26.8 vs 25.6 us).
    const int ntiles = n_codes >> 5;
    int nsplit = 1;
    {
        double best = 1e300;
        for (int s = 1; s <= 16; s *= 2) {
            if (s > 1 && ntiles / s < 16) break;
            const double cost = ((double)blocks * s / 512.0 + 0.5) * ((double)(ntiles + s - 1) / s + 10.0);
            if (cost < best) { best = cost; nsplit = s; }
        }
    }
    if (g_vq_split >= 1) nsplit = g_vq_split;  // "vq_split" option: force (tests / A-B)
    if (nsplit > ntiles) nsplit = ntiles;
    if (VAR == 2)  // the staged fragments (1 KiB per tile) + ee must fit the CU's LDS
        while ((ntiles + nsplit - 1) / nsplit * (1024 + 128) + 64 > 160 * 1024) nsplit *= 2;
    const int tiles_per = (ntiles + nsplit - 1) / nsplit;
    const int lds = (cosine ? 0 : (tiles_per * 32 + 3) / 4 * 4 * 4) + (VAR == 2 ? tiles_per * 1024 : 0);
    if (lds > 65536) {
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(vq_argmin_kernel<false, MODE, VAR>), lds)) return rc;
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(vq_argmin_kernel<true, MODE, VAR>), lds)) return rc;
    }
    const dim3 grid((unsigned)blocks, nsplit);
    if (nsplit == 1) {
        hipLaunchKernelGGL((vq_argmin_kernel<false, MODE, VAR>), grid, dim3(256), lds, stream, z, packed, ee, n, n_codes, ids);
        OT_LAUNCH_CHECK("vq_argmin");
        return OMNITOK_OK;
    }
    if (int rc = device_fill_u32(ids, 0xFFFFFFFFu, n * 2, stream)) return rc;
    hipLaunchKernelGGL((vq_argmin_kernel<true, MODE, VAR>), grid, dim3(256), lds, stream, z, packed, ee, n, n_codes, ids);
    OT_LAUNCH_CHECK("vq_argmin");
    hipLaunchKernelGGL(vq_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, n);
    OT_LAUNCH_CHECK("vq_finalize");
    return OMNITOK_OK;
}

template <int MODE>
static int launch_vq(const float *z, const float *packed, const float *ee, int64_t n, int n_codes, int64_t *ids,
                     hipStream_t stream) {
    if constexpr (MODE == VQ_CODEBOOK) {
        if (g_vq_variant == 0) return launch_vq_var<MODE, 0>(z, packed, ee, n, n_codes, ids, stream);
        if (g_vq_variant == 2) return launch_vq_var<MODE, 2>(z, packed, ee, n, n_codes, ids, stream);
        return launch_vq_var<MODE, 1>(z, packed, ee, n, n_codes, ids, stream);
    }
    return launch_vq_var<MODE, 0>(z, packed, ee, n, n_codes, ids, stream);
}

extern "C" int omnitok_vq_argmin(const float *z, const float *packed, const float *ee, int64_t n, int n_codes,
                                 int64_t *ids, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(z && packed && ee && ids, "vq_argmin: null pointer");
    OT_CHECK_ARG(n_codes % 32 == 0 && n_codes > 0 && n_codes <= 32768, "vq_argmin: n_codes=%d unsupported", n_codes);
    OT_CHECK_ARG(aligned16(z) && aligned16(packed) && aligned16(ee), "vq_argmin: unaligned");
    return launch_vq<VQ_CODEBOOK>(z, packed, ee, n, n_codes, ids, stream);
}

// screen[n_codes * 8 + 4] floats: 32 bytes per code of fp16 screening fragments (per tile of 32 codes 64 lanes x 16 B in MFMA
// A-fragment order: lanes 0..31 the code's 8 fp16 coordinates, lanes 32..63 its ee as two fp16 halves) followed by
// {max |E_c|_2, max ee_c, max |E_ck|, 0}
extern "C" int omnitok_vq_screen_prepare(const float *codebook, const float *ee, int n_codes, int cdim, float *screen,
                                         omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(codebook && ee && screen, "vq_screen_prepare: null pointer");
    OT_CHECK_ARG(cdim == 8 && n_codes % 32 == 0 && n_codes > 0 && aligned16(screen), "vq_screen_prepare: need cdim == 8, n_codes %% 32 == 0");
    unsigned *consts = reinterpret_cast<unsigned *>(screen + (int64_t)n_codes * 8);
    if (int rc = device_fill_u32(consts, 0u, 4, stream)) return rc;
    hipLaunchKernelGGL(vq_screen_prepare_kernel, dim3((n_codes * 2 + 255) / 256), dim3(256), 0, stream, codebook, ee, n_codes,
                       reinterpret_cast<u32x4 *>(screen), consts);
    OT_LAUNCH_CHECK("vq_screen_prepare");
    return OMNITOK_OK;
}

extern "C" int omnitok_vq_argmin_screened(const float *z, const float *packed, const float *ee, const float *screen, int64_t n,
                                          int n_codes, int64_t *ids, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(z && packed && ee && screen && ids, "vq_argmin_screened: null pointer");
    OT_CHECK_ARG(n_codes % 32 == 0 && n_codes > 0 && n_codes <= 32768, "vq_argmin_screened: n_codes=%d unsupported", n_codes);
    OT_CHECK_ARG(aligned16(z) && aligned16(packed) && aligned16(ee) && aligned16(screen), "vq_argmin_screened: unaligned");
    const int64_t blocks = (n + VQ_ROWS_PER_BLOCK - 1) / VQ_ROWS_PER_BLOCK;
    const int ntiles = n_codes >> 5;
    // splits of the code range, from the sweep in profiles/r05_vq_screen.txt (C3 163 840 x 8192: 378 / 222 / 130 / 168 / 243 us
    // at 1 / 2 / 4 / 8 / 16 splits; C5 69 632 x 16 384: best at 4-8; one clip / one image: best at 16): about 64 tiles per
    // workgroup, more splits only while the launch has fewer than 1024 workgroups and a split keeps >= 16 tiles
    int nsplit = 1;
    while (ntiles / nsplit > 64 && nsplit < 16) nsplit *= 2;
    while (blocks * nsplit < 1024 && nsplit < 16 && ntiles / (nsplit * 2) >= 16) nsplit *= 2;
    if (g_vq_screen_split >= 1) nsplit = g_vq_screen_split;
    if (nsplit > ntiles) nsplit = ntiles;
    const int tiles_per = (ntiles + nsplit - 1) / nsplit;
    const int lds = (tiles_per * 32 + 3) / 4 * 4 * 4;
    if (lds > 65536) {
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(vq_screen_kernel<false>), lds)) return rc;
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(vq_screen_kernel<true>), lds)) return rc;
    }
    const u32x4 *frag = reinterpret_cast<const u32x4 *>(screen);
    const float *consts = screen + (int64_t)n_codes * 8;
    const dim3 grid((unsigned)blocks, nsplit);
    if (nsplit == 1) {
        hipLaunchKernelGGL((vq_screen_kernel<false>), grid, dim3(256), lds, stream, z, packed, ee, frag, consts, n, n_codes, ids);
        OT_LAUNCH_CHECK("vq_screen");
        return OMNITOK_OK;
    }
    if (int rc = device_fill_u32(ids, 0xFFFFFFFFu, n * 2, stream)) return rc;
    hipLaunchKernelGGL((vq_screen_kernel<true>), grid, dim3(256), lds, stream, z, packed, ee, frag, consts, n, n_codes, ids);
    OT_LAUNCH_CHECK("vq_screen");
    hipLaunchKernelGGL(vq_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, n);
    OT_LAUNCH_CHECK("vq_finalize");
    return OMNITOK_OK;
}

extern "C" int omnitok_vq_argmax_cos(const float *z, const float *packed, int64_t n, int n_codes, int64_t *ids,
                                     omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(z && packed && ids, "vq_argmax_cos: null pointer");
    OT_CHECK_ARG(n_codes % 32 == 0 && n_codes > 0 && n_codes <= 32768, "vq_argmax_cos: n_codes=%d unsupported", n_codes);
    OT_CHECK_ARG(aligned16(z) && aligned16(packed), "vq_argmax_cos: unaligned");
    return launch_vq<VQ_COS>(z, packed, nullptr, n, n_codes, ids, stream);
}

extern "C" int omnitok_vq_argmin_cdist(const float *z, const float *packed, const float *ee, int64_t n, int n_codes,
                                       int64_t *ids, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(z && packed && ee && ids, "vq_argmin_cdist: null pointer");
    OT_CHECK_ARG(n_codes % 32 == 0 && n_codes > 0 && n_codes <= 32768, "vq_argmin_cdist: n_codes=%d unsupported",
                 n_codes);
    OT_CHECK_ARG(aligned16(z) && aligned16(packed) && aligned16(ee), "vq_argmin_cdist: unaligned");
    return launch_vq<VQ_EUCLID>(z, packed, ee, n, n_codes, ids, stream);
}

extern "C" int omnitok_dequant_table(const float *codebook, int n_codes, int cdim, const float *w, const float *b,
                                     float *table, int D, int64_t *scratch_ids, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(codebook && w && b && table && scratch_ids, "dequant_table: null pointer");
    OT_CHECK_ARG(cdim == 8 && D % 4 == 0 && n_codes > 0, "dequant_table: need cdim == 8, D %% 4 == 0, n_codes > 0");
    hipLaunchKernelGGL(iota_i64_kernel, dim3((n_codes + 255) / 256), dim3(256), 0, stream, scratch_ids, n_codes);
    OT_LAUNCH_CHECK("iota");
    const int64_t total = (int64_t)n_codes * (D / 4);
    hipLaunchKernelGGL(dequant_post_vq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, scratch_ids,
                       codebook, n_codes, w, b, table, (int64_t)n_codes, D, nullptr);
    OT_LAUNCH_CHECK("dequant_table");
    return OMNITOK_OK;
}

extern "C" int omnitok_gather_rows(const int64_t *ids, const float *table, int n_codes, float *tok, int64_t n, int D,
                                   int *err_flag, omnitok_stream_t stream_) {
    return omnitok_gather_rows_transposed(ids, table, n_codes, tok, n, 0, 0, D, err_flag, stream_);
}

extern "C" int omnitok_gather_rows_transposed(const int64_t *ids, const float *table, int n_codes, float *tok, int64_t n,
                                              int a, int c, int D, int *err_flag, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(ids && table && tok, "gather_rows: null pointer");
    OT_CHECK_ARG(D % 4 == 0 && aligned16(table) && aligned16(tok), "gather_rows: need D %% 4 == 0 and 16-byte alignment");
    OT_CHECK_ARG((a == 0 && c == 0) || (a > 0 && c > 0 && n % ((int64_t)a * c) == 0),
                 "gather_rows: %lld rows are not whole [a=%d, c=%d] groups", (long long)n, a, c);
    if (n == 0) return OMNITOK_OK;
    const int64_t total = n * (D / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ids,
                       reinterpret_cast<const f32x4 *>(table), n_codes, reinterpret_cast<f32x4 *>(tok), n, D / 4,
                       err_flag, a, c);
    OT_LAUNCH_CHECK("gather_rows");
    return OMNITOK_OK;
}

extern "C" int omnitok_dequant_post_vq(const int64_t *ids, const float *codebook, int n_codes, int cdim,
                                       const float *w, const float *b, float *tok, int64_t n, int D, int *err_flag,
                                       omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(ids && codebook && w && b && tok, "dequant_post_vq: null pointer");
    OT_CHECK_ARG(cdim == 8 && D % 4 == 0, "dequant_post_vq: need cdim == 8 and D %% 4 == 0");
    if (n == 0) return OMNITOK_OK;
    const int64_t total = n * (D / 4);
    hipLaunchKernelGGL(dequant_post_vq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ids,
                       codebook, n_codes, w, b, tok, n, D, err_flag);
    OT_LAUNCH_CHECK("dequant_post_vq");
    return OMNITOK_OK;
}

extern "C" int omnitok_vq_embed_st(const int64_t *ids, const float *z, const float *codebook, int cdim, int64_t B,
                                   int64_t thw, float *emb, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(ids && z && codebook && emb && cdim == 8, "vq_embed_st: bad arguments");
    const int64_t total = B * thw;
    if (total == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(vq_embed_st_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ids, z,
                       codebook, B, thw, emb);
    OT_LAUNCH_CHECK("vq_embed_st");
    return OMNITOK_OK;
}

extern "C" int omnitok_vae_sample(const float *x, const float *w, const float *b, const float *noise, float *z,
                                  float *moments, int64_t B, int64_t thw, int D, int cdim,
                                  omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t n = B * thw;
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(x && w && b && z, "vae_sample: null pointer");
    OT_CHECK_ARG(cdim == 8, "vae_sample: codebook_dim=%d (kernels are built for 8)", cdim);
    OT_CHECK_ARG(D % 64 == 0, "vae_sample: D %% 64 != 0");
    const int64_t threads = n * 16;
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, w, b,
                       noise, z, moments, n, thw, D);
    OT_LAUNCH_CHECK("vae_sample");
    return OMNITOK_OK;
}

extern "C" int omnitok_post_vq(const float *z, int channel_first, int64_t B, int64_t thw, int cdim, const float *w,
                               const float *b, float *tok, int D, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t n = B * thw;
    if (n == 0) return OMNITOK_OK;
    OT_CHECK_ARG(z && w && b && tok, "post_vq: null pointer");
    OT_CHECK_ARG(cdim == 8 && D % 4 == 0, "post_vq: need cdim == 8 and D %% 4 == 0");
    const int64_t total = n * (D / 4);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (channel_first)
        hipLaunchKernelGGL(post_vq_latent_kernel<true>, grid, dim3(256), 0, stream, z, thw, w, b, tok, n, D);
    else
        hipLaunchKernelGGL(post_vq_latent_kernel<false>, grid, dim3(256), 0, stream, z, thw, w, b, tok, n, D);
    OT_LAUNCH_CHECK("post_vq");
    return OMNITOK_OK;
}
