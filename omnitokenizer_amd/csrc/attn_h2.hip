// Spatial attention on the fp16 matrix cores from 2-way splits of every operand ("h2" attention):
//   * attn_pack          -- RoPE(2-D) + l2norm + q/k scale (reference attention.py:417-437), then Q, K and V are
//                           written as fp16 hi|lo planes in MFMA-fragment order
//   * attn_spatial_h2    -- full attention over N tokens, flash style (attention.py:439-483)
//
// Why: after the GEMMs moved to the fp16 pipe (gemm_h2.hip) the fp32-MFMA attention kernel was 21 % of the
// C3 step at 110 TF.  The same operand split applies here with rigorous ranges for free: q and k are
// l2-normalised (|q_d| <= 8 max|q_scale|, |k_d| <= max|k_scale|), P = exp(s - m) is in (0, 1], and V is bounded
// per clip by the row-statistics pass (|V_j| <= ||x_row|| ||Wv_j||).
//
// Arithmetic.  Every operand element x (scaled by an exact power of two so that the bound maps to (2^14, 2^15])
// is split  x = hi + lo + r,  hi = fp16(x),  lo = fp16(x - hi),  |r| <= 2^-22 |x|;  each product is accumulated
// in fp32 as lo.hi' + hi.lo' + hi.hi' (three v_mfma_f32_32x32x16_f16; the dropped lo.lo' and r terms are below
// 2^-21 |x y|).  The softmax itself (max, exp2, sum, rescale) is the fp32 code of attn_spatial.hip; P is scaled
// by 2^14 before it is split (the factor cancels in O / l).  Error class: tests/test_gpu_attn_h2.py puts the
// result within the fp32-MFMA kernel's distance of an fp64 reference.
//
// MFMA formulation: transposed like attn_spatial.hip so that each lane owns one query column --
//     S^T[key][q] = K . Q^T      A = K fragment, B = Q fragment   (4 k-steps of 16 channels, 12 MFMA / 32x32)
//     O^T[d][q]   = V^T . P^T    A = V fragment, B = P fragment   (2 k-steps of 16 keys x 2 d-halves, 12 MFMA)
// In the 32x32 C/D layout lane l holds S^T rows 8 i + 4 (l >> 5) + r (i, r in 0..3) of column q = l & 31, and
// the B operand of 32x32x16 wants from lane l eight consecutive k.  The key order inside a 16-key step is ours
// to choose (the contraction runs over keys), so k-slot 8 h + 4 ii + r of step j is DEFINED to be key
// 16 j + 8 ii + 4 h + r: P goes from the S^T accumulators to the second MFMA without any cross-lane move, and
// the V planes are stored in exactly that order by attn_pack.
//
// Packed layouts (all in units of one 32-token block of one (sequence, head): 8 KiB = the fp32 size):
//   Q, K:  [plane 2][k-step 4][h 2][token 32][8 halfs]   element = x[token][16 ks + 8 h + e]
//   V:     [plane 2][j 2][d-half 2][h 2][d 32][8 halfs]  element = v[16 j + 8 (e >> 2) + 4 h + (e & 3)][32 mt + d]
// so that every fragment is one conflict-free ds_read_b128 per lane (lanes 0-31 and 32-63 read two contiguous
// 512-byte runs) and a K/V tile is staged by a linear copy.
#include "h2_common.h"
#include "planes.h"

namespace omnitok {

constexpr int PK_CHUNK = 528;  // LDS stride of a 512-byte run in attn_pack (16 B pad: conflict-free b64 stores)
constexpr float P_SHIFT = 14.0f;  // P is scaled by 2^14 before the fp16 split

struct PackParams {
    const float *q; int64_t ldq;
    const float *k; const float *v; int64_t ldkv;
    int n_tokens, heads, nblk;       // tokens per sequence, heads, 32-token blocks per sequence
    const float *cosT, *sinT, *q_scale, *k_scale;
    float scale;                     // SDPA scale folded into q (reference attention.py:431-437)
    float sq, sk;                    // power-of-two operand scales of q and k
    float v_bound; const float *v_bound_dev; int v_bound_stride; int64_t rows_per_clip;
    unsigned char *qp, *kp, *vp;
};

// One workgroup packs one 32-token block of one head (Q, K and V).
__global__ __launch_bounds__(256) void attn_pack_kernel(PackParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 16 * PK_CHUNK];
    const int tid = threadIdx.x;
    const int64_t unit = blockIdx.x;  // (seq, head, blk) with blk fastest
    const int blk = (int)(unit % p.nblk);
    const int head = (int)((unit / p.nblk) % p.heads);
    const int64_t seq = unit / ((int64_t)p.nblk * p.heads);
    const int64_t row0 = seq * p.n_tokens + (int64_t)blk * 32;

    // ---- Q and K: one 16-lane row per token, 4 consecutive channels per lane (RoPE pairs are lane-local,
    //      the l2 norm is a DPP all-reduce) -- the arithmetic of qk_prep_kernel, bit for bit ---------------
    {
        const int l16 = tid & 15, tk = tid >> 4;
        const int ks = l16 >> 2, h = (l16 >> 1) & 1, e0 = 4 * (l16 & 1);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int tok = pass * 16 + tk;
            const int64_t row = row0 + tok;
            const int n = (int)(row % p.n_tokens);
            float c0 = 1.f, c1 = 1.f, s0 = 0.f, s1 = 0.f;
            if (p.cosT) {
                c0 = p.cosT[n * 32 + 2 * l16]; c1 = p.cosT[n * 32 + 2 * l16 + 1];
                s0 = p.sinT[n * 32 + 2 * l16]; s1 = p.sinT[n * 32 + 2 * l16 + 1];
            }
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const float *base = which == 0 ? p.q + row * p.ldq : p.k + row * p.ldkv;
                f32x4 v = *reinterpret_cast<const f32x4 *>(base + head * 64 + l16 * 4);
                if (p.cosT) {  // (a + ib)(c + is) = (ac - bs) + i(as + bc), reference attention.py:65-69
                    const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
                    v[0] = a0 * c0 - b0 * s0; v[1] = a0 * s0 + b0 * c0;
                    v[2] = a1 * c1 - b1 * s1; v[3] = a1 * s1 + b1 * c1;
                }
                float ss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                ss = row16_allsum(ss);
                const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps, attention.py:24-25
                const f32x4 sc = *reinterpret_cast<const f32x4 *>((which == 0 ? p.q_scale : p.k_scale) + l16 * 4);
                const float mul = which == 0 ? p.scale : 1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * inv * sc[e] * mul;
                v *= which == 0 ? p.sq : p.sk;  // exact (power of two)
                const f16x4 hh = __builtin_convertvector(v, f16x4);
                const f16x4 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f32x4), f16x4);
                unsigned char *dst = lds + which * 16 * PK_CHUNK + (ks * 2 + h) * PK_CHUNK + tok * 16 + e0 * 2;
                *reinterpret_cast<u32x2 *>(dst) = __builtin_bit_cast(u32x2, hh);
                *reinterpret_cast<u32x2 *>(dst + 8 * PK_CHUNK) = __builtin_bit_cast(u32x2, ll);
            }
        }
    }
    // ---- V: lane = channel d, each wave takes two key quads (j, ii, h): four keys 16 j + 8 ii + 4 h + r --------
    //      (skipped when the q|k|v GEMM already wrote the V planes from its epilogue: omnitok_gemm_h2_vpack)
    if (p.vp) {
        const int lane = tid & 63, wave = tid >> 6;
        const int mt = lane >> 5, dd = lane & 31;
        float bound = p.v_bound;
        if (p.v_bound_dev) bound *= p.v_bound_dev[(int64_t)p.v_bound_stride * (row0 / p.rows_per_clip)];
        const float sv = h2_scale_of_bound(bound);
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int quad = wave * 2 + qi;
            const int j = quad >> 2, ii = (quad >> 1) & 1, h = quad & 1;
            const int kl0 = 16 * j + 8 * ii + 4 * h;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = p.v[(row0 + kl0 + r) * p.ldkv + head * 64 + lane];
            v *= sv;
            const f16x4 hh = __builtin_convertvector(v, f16x4);
            const f16x4 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f32x4), f16x4);
            unsigned char *dst = lds + 2 * 16 * PK_CHUNK + ((j * 2 + mt) * 2 + h) * PK_CHUNK + dd * 16 + ii * 8;
            *reinterpret_cast<u32x2 *>(dst) = __builtin_bit_cast(u32x2, hh);
            *reinterpret_cast<u32x2 *>(dst + 8 * PK_CHUNK) = __builtin_bit_cast(u32x2, ll);
        }
    }
    __syncthreads();
    // ---- linear 8 KiB stores of the three blocks --------------------------------------------------------
    unsigned char *outs[3] = {p.qp, p.kp, p.vp};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t == 2 && !p.vp) break;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = tid + 256 * i;  // 16-byte slot 0..511
            const u32x4 w = *reinterpret_cast<const u32x4 *>(lds + (t * 16 + (s >> 5)) * PK_CHUNK + (s & 31) * 16);
            *reinterpret_cast<u32x4 *>(outs[t] + unit * 8192 + (int64_t)s * 16) = w;
        }
    }
}

// -------------------------------------------------------------------------------------------
// Attention.  Workgroup = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries.
// K/V tiles of 64 keys (16 KiB + 16 KiB, already in fragment order) are staged global -> registers ->
// LDS, double-buffered: the next tile's loads are in flight during the current tile's 48 MFMAs.
// -------------------------------------------------------------------------------------------
constexpr int AH_TILE_BYTES = 32768;            // K (2 blocks) | V (2 blocks)
constexpr int AH_LDS_BYTES = 2 * AH_TILE_BYTES;

struct AttnH2Params {
    const unsigned char *qp, *kp, *vp;
    float *out; int64_t ldo;
    int N, heads, nqb, ngrp_real;
    float s_unscale;  // 1 / (sq sk): undoes the q/k operand scales (exact power of two)
    float v_bound; const float *v_bound_dev; int v_bound_stride; int seq_per_clip;
    const float *bias_table;  // [(2gh-1)*(2gw-1), heads] or null
    int gh, gw;
    int lg_gw;   // log2(gw) when the 64-query kernels stage the head's bias table in LDS (gw a power of two >= 8), else -1
    // optional: the output as fp16 hi|lo planes, the A operand of the to_out GEMM (gemm_pl.h), instead of fp32 rows;
    // scaled per clip by the power of two of the V bound (|O| <= max |V|), whose inverse goes to out_scale[row]
    unsigned char *out_planes;
    float *out_scale;
};

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_spatial_h2_kernel(AttnH2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    // XCD-aware mapping as attn_spatial_kernel: all query blocks of one (sequence, head) on one XCD
    const int nqb = p.nqb;
    const int b = blockIdx.x, jb = b >> 3;
    const int grp = (jb / nqb) * 8 + (b & 7);
    if (grp >= p.ngrp_real) return;
    const int qb = jb % nqb, head = grp % p.heads, seq = grp / p.heads;
    const int nblk = p.N / 32;
    const int64_t unit0 = ((int64_t)seq * p.heads + head) * nblk;  // first 32-token block of this (sequence, head)
    const bool wave_active = qb * 128 + wave * 32 < p.N;
    const int qblk = wave_active ? qb * 4 + wave : 0;
    const int q_local = qblk * 32 + r32;  // this lane's query

    // Q fragments (B operand): plane, k-step -> 8 halfs of query r32, channels 16 ks + 8 hi + (0..7)
    u32x4 qf[2][4];
    {
        const unsigned char *qb_ = p.qp + (unit0 + qblk) * 8192 + hi * 512 + r32 * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pl][ks] = *reinterpret_cast<const u32x4 *>(qb_ + (pl * 4 + ks) * 1024);
    }
    int qy = 0, qx = 0;
    const float *btab = nullptr;
    if constexpr (HAS_BIAS) {
        qy = q_local / p.gw; qx = q_local % p.gw;
        btab = p.bias_table + head;
    }

    // loader: a tile is 2 x 16 KiB of contiguous global memory; thread -> 16-byte slots tid + 256 i
    const unsigned char *kg = p.kp + unit0 * 8192 + tid * 16;
    const unsigned char *vg = p.vp + unit0 * 8192 + tid * 16;
    u32x4 rk[4], rv[4];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rk[i] = *reinterpret_cast<const u32x4 *>(kg + (int64_t)t * 16384 + i * 4096);
            rv[i] = *reinterpret_cast<const u32x4 *>(vg + (int64_t)t * 16384 + i * 4096);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char *s = smem_h2 + buf * AH_TILE_BYTES + tid * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4 *>(s + i * 4096) = rk[i];
            *reinterpret_cast<u32x4 *>(s + 16384 + i * 4096) = rv[i];
        }
    };

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    // logits = s_unscale * acc (+ bias); softmax in exp2 form
    const float cs = HAS_BIAS ? 1.44269504088896340736f : p.s_unscale * 1.44269504088896340736f;

    const int ntiles = p.N / 64;
    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const unsigned char *Ks = smem_h2 + buf * AH_TILE_BYTES + hi * 512 + r32 * 16;
        const unsigned char *Vs = Ks + 16384;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // ---- S^T = K . Q^T for 32 keys x 32 queries ---------------------------------------
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(Ks + sub * 8192 + ks * 1024));
                const f16x8 kl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(Ks + sub * 8192 + (4 + ks) * 1024));
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, __builtin_bit_cast(f16x8, qf[0][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[1][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[0][ks]), st, 0, 0, 0);
            }
            if constexpr (HAS_BIAS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * 64 + sub * 32 + mfma32_row(r, hi);
                    const int ky = kv / p.gw, kx = kv % p.gw;
                    st[r] = fmaf(st[r], p.s_unscale,
                                 btab[((qy - ky + p.gh - 1) * (2 * p.gw - 1) + (qx - kx + p.gw - 1)) * p.heads]);
                }
            }
            // ---- online softmax (per-lane query), in units of the raw accumulators -----------------
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = fmaxf(mx, swap32(mx));
            const float m_new = fmaxf(m_run, mx);
            const float mc = fmaf(m_new, cs, -P_SHIFT);  // exp2(s cs - mc) = 2^14 exp(logit - max)
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], cs, -mc));
                ps += st[r];
            }
            ps += swap32(ps);
            if (__any(m_new != m_run)) {  // wave-uniform; alpha == 1 for every lane otherwise
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);  // 0 on the first tile
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            }
            l_run += ps;
            m_run = m_new;
            // ---- split P (2 k-steps of 16 keys: accumulators 8 j .. 8 j + 7) and O^T += V^T . P^T ------
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pa[e] = st[8 * j + e];
                    pb[e] = st[8 * j + 4 + e];
                }
                const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                const f16x8 ph = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                const f16x8 pl = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 vh = __builtin_bit_cast(
                        f16x8, *reinterpret_cast<const u32x4 *>(Vs + sub * 8192 + (j * 2 + mt) * 1024));
                    const f16x8 vl = __builtin_bit_cast(
                        f16x8, *reinterpret_cast<const u32x4 *>(Vs + sub * 8192 + 4096 + (j * 2 + mt) * 1024));
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, ot[mt], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / (l sv); lane owns q, d = mt*32 + 8g + 4hi + (0..3) -------------
    if (!wave_active) return;
    float bound = p.v_bound;
    if (p.v_bound_dev) bound *= p.v_bound_dev[(int64_t)p.v_bound_stride * (seq / p.seq_per_clip)];
    const float inv_l = 1.0f / (l_run * h2_scale_of_bound(bound));  // the scale is a power of two: exact
    if (p.out_planes) {
        const float so = h2_scale_of_bound(bound);
        const int64_t row = (int64_t)seq * p.N + q_local;
        if (head == 0 && hi == 0) p.out_scale[row] = 1.0f / so;
#pragma unroll
        for (int d = 0; d < 2; ++d) pl_store_ot_block(ot[d], inv_l * so, p.out_planes, row, head * 2 + d, p.heads * 2, hi);
        return;
    }
    float *orow = p.out + ((int64_t)seq * p.N + q_local) * p.ldo + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[d][g * 4 + e] * inv_l;
            *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
        }
}


// -------------------------------------------------------------------------------------------
// Variant 2 ("attn_h2_variant" 1, the default): the same arithmetic, restructured around what the first kernel
// measured (matrix pipe ~35 % busy: every wave alternates MFMA bursts with ~140-instruction softmax bursts):
//   * K/V tiles go global -> LDS directly (global_load_lds_dwordx4: the packed layout is lane-linear), no
//     staging registers and no ds_write pass;
//   * inside a 64-key tile the two 32-key sub-blocks are software-pipelined: S^T of sub-block b is issued
//     before the softmax of sub-block a, and the P.V MFMAs of a before the softmax of b, so the VALU work of one
//     sub-block sits in the shadow of the other's MFMAs within ONE wave (regions without branches);
//   * the two half-wave reductions use v_permlane32_swap instead of ds_bpermute.
// -------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glob_void_t;

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_spatial_h2p_kernel(AttnH2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nqb = p.nqb;
    const int b = blockIdx.x, jb = b >> 3;
    const int grp = (jb / nqb) * 8 + (b & 7);
    if (grp >= p.ngrp_real) return;
    const int qb = jb % nqb, head = grp % p.heads, seq = grp / p.heads;
    const int nblk = p.N / 32;
    const int64_t unit0 = ((int64_t)seq * p.heads + head) * nblk;
    const bool wave_active = qb * 128 + wave * 32 < p.N;
    const int qblk = wave_active ? qb * 4 + wave : 0;
    const int q_local = qblk * 32 + r32;

    u32x4 qf[2][4];
    {
        const unsigned char *qb_ = p.qp + (unit0 + qblk) * 8192 + hi * 512 + r32 * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pl][ks] = *reinterpret_cast<const u32x4 *>(qb_ + (pl * 4 + ks) * 1024);
    }
    int qy = 0, qx = 0;
    const float *btab = nullptr;
    if constexpr (HAS_BIAS) {
        qy = q_local / p.gw; qx = q_local % p.gw;
        btab = p.bias_table + head;
    }

    // tile t = 16 KiB of K + 16 KiB of V, contiguous in global memory and lane-linear in LDS: wave w moves the
    // 1 KiB chunks 4 i + w (i = 0..3) of each half
    const unsigned char *kg = p.kp + unit0 * 8192 + wave * 1024 + lane * 16;
    const unsigned char *vg = p.vp + unit0 * 8192 + wave * 1024 + lane * 16;
    auto dma = [&](int t, int buf) {
        unsigned char *s = smem_h2 + buf * AH_TILE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glob_void_t *)(kg + (int64_t)t * 16384 + i * 4096),
                                             (lds_void_t *)(s + i * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glob_void_t *)(vg + (int64_t)t * 16384 + i * 4096),
                                             (lds_void_t *)(s + 16384 + i * 4096), 16, 0, 0);
        }
    };

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float cs = HAS_BIAS ? 1.44269504088896340736f : p.s_unscale * 1.44269504088896340736f;

    const int ntiles = p.N / 64;
    dma(0, 0);
    // the LDS-DMA loads are tracked by vmcnt only: drain them explicitly before the barrier that publishes the tile
    // (hipcc adds this wait itself when it knows a global_load_lds is outstanding; stated here so that the
    // ordering does not depend on that analysis)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) dma(t + 1, buf ^ 1);
        const unsigned char *Ks = smem_h2 + buf * AH_TILE_BYTES + hi * 512 + r32 * 16;
        const unsigned char *Vs = Ks + 16384;

        auto qk = [&](int sub) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(Ks + sub * 8192 + ks * 1024));
                const f16x8 kl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(Ks + sub * 8192 + (4 + ks) * 1024));
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, __builtin_bit_cast(f16x8, qf[0][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[1][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[0][ks]), st, 0, 0, 0);
            }
            if constexpr (HAS_BIAS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * 64 + sub * 32 + mfma32_row(r, hi);
                    const int ky = kv / p.gw, kx = kv % p.gw;
                    st[r] = fmaf(st[r], p.s_unscale,
                                 btab[((qy - ky + p.gh - 1) * (2 * p.gw - 1) + (qx - kx + p.gw - 1)) * p.heads]);
                }
            }
            return st;
        };
        // online softmax of one sub-block: updates m_run / l_run, returns alpha (1 when the maximum did not grow)
        // and P as hi|lo fp16 fragments of the two 16-key steps
        auto softmax = [&](f32x16 &st, f16x8 (&ph)[2], f16x8 (&pl)[2]) {
            float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
            mx = halves_max(fmaxf(mx, st[15]));
            const float m_new = fmaxf(m_run, mx);
            const float mc = fmaf(m_new, cs, -P_SHIFT);
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], cs, -mc));
                ps += st[r];
            }
            ps = halves_sum(ps);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);  // 1 if unchanged, 0 on the first block
            l_run = fmaf(l_run, alpha, ps);
            m_run = m_new;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pa[e] = st[8 * j + e];
                    pb[e] = st[8 * j + 4 + e];
                }
                const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                ph[j] = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                pl[j] = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            return alpha;
        };
        auto rescale = [&](float alpha) {
            if (__any(alpha != 1.0f)) {  // wave-uniform branch; exact: alpha == 1 for every lane otherwise
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            }
        };
        auto pv = [&](int sub, const f16x8 (&ph)[2], const f16x8 (&pl)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 vh = __builtin_bit_cast(
                        f16x8, *reinterpret_cast<const u32x4 *>(Vs + sub * 8192 + (j * 2 + mt) * 1024));
                    const f16x8 vl = __builtin_bit_cast(
                        f16x8, *reinterpret_cast<const u32x4 *>(Vs + sub * 8192 + 4096 + (j * 2 + mt) * 1024));
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[j], ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[j], ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[j], ot[mt], 0, 0, 0);
                }
        };

        f16x8 pha[2], pla[2], phb[2], plb[2];
        f32x16 sa = qk(0);
        f32x16 sb = qk(1);                       // S^T(b) MFMAs ...
        const float alpha_a = softmax(sa, pha, pla);   // ... beside the softmax of a
        rescale(alpha_a);
        pv(0, pha, pla);                         // P.V(a) MFMAs ...
        const float alpha_b = softmax(sb, phb, plb);   // ... beside the softmax of b
        rescale(alpha_b);
        pv(1, phb, plb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t + 1 has landed (see the prologue)
        __syncthreads();
    }

    if (!wave_active) return;
    float bound = p.v_bound;
    if (p.v_bound_dev) bound *= p.v_bound_dev[(int64_t)p.v_bound_stride * (seq / p.seq_per_clip)];
    const float inv_l = 1.0f / (l_run * h2_scale_of_bound(bound));
    if (p.out_planes) {
        const float so = h2_scale_of_bound(bound);
        const int64_t row = (int64_t)seq * p.N + q_local;
        if (head == 0 && hi == 0) p.out_scale[row] = 1.0f / so;
#pragma unroll
        for (int d = 0; d < 2; ++d) pl_store_ot_block(ot[d], inv_l * so, p.out_planes, row, head * 2 + d, p.heads * 2, hi);
        return;
    }
    float *orow = p.out + ((int64_t)seq * p.N + q_local) * p.ldo + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[d][g * 4 + e] * inv_l;
            *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
        }
}


// -------------------------------------------------------------------------------------------
// Variants 2 and 3 ("attn_h2_variant" 2 / 3): the arithmetic and the operand formats of the kernel above, with the
// instruction stream of a tile arranged around what its ISA showed (profiles/r04_attn_h2_isa.txt).  hipcc emitted the K
// and V fragment reads right in front of their uses -- `2 x ds_read_b128, s_waitcnt lgkmcnt(0), 3 x MFMA` twelve times per
// sub-block -- so every MFMA triple started with an exposed LDS round trip, and it kept the softmax of sub-block a
// AFTER the S^T MFMAs of sub-block b in program order (an in-order wave then runs them one after the other whatever the
// source intended).  Here
//   * all fragment reads of a phase are issued as one batch BEFORE the phase that precedes their use (the K fragments
//     of both sub-blocks at the top of the tile, the V fragments of a sub-block before its softmax), pinned with
//     sched_barrier fences: every MFMA chain finds its operands in registers and issues back to back;
//   * SUBS = 2 (variant 2): 64-key tiles, two workgroups per CU as before.  SUBS = 1 (variant 3): 32-key tiles, 32 KiB of
//     LDS per workgroup and <= 168 registers: three workgroups per CU, so that a SIMD has three waves in different phases
//     (one in its MFMA chain, the others in softmax / waiting on the barrier).
// -------------------------------------------------------------------------------------------
// DBG (measurement builds behind "attn_h2_dbg", wrong results): 1 no softmax arithmetic (P = S), 2 no S^T MFMAs, 4 no P.V MFMAs,
// 8 no barrier / DMA wait in the tile loop, 16 no DMA in the tile loop, 32 no P split (lo = hi)
template <bool HAS_BIAS, int SUBS, int DBG = 0, int MINB = (SUBS == 1 ? 3 : 2)>
__global__ __launch_bounds__(256, MINB) void attn_spatial_h2x_kernel(AttnH2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    constexpr int TILE = SUBS * 16384;  // K (SUBS blocks of 8 KiB) | V (SUBS blocks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nqb = p.nqb;
    const int b = blockIdx.x, jb = b >> 3;
    const int grp = (jb / nqb) * 8 + (b & 7);
    if (grp >= p.ngrp_real) return;
    const int qb = jb % nqb, head = grp % p.heads, seq = grp / p.heads;
    const int nblk = p.N / 32;
    const int64_t unit0 = ((int64_t)seq * p.heads + head) * nblk;
    const bool wave_active = qb * 128 + wave * 32 < p.N;
    const int qblk = wave_active ? qb * 4 + wave : 0;
    const int q_local = qblk * 32 + r32;

    u32x4 qf[2][4];
    {
        const unsigned char *qb_ = p.qp + (unit0 + qblk) * 8192 + hi * 512 + r32 * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pl][ks] = *reinterpret_cast<const u32x4 *>(qb_ + (pl * 4 + ks) * 1024);
    }
    int qy = 0, qx = 0;
    const float *btab = nullptr;
    if constexpr (HAS_BIAS) {
        qy = q_local / p.gw; qx = q_local % p.gw;
        btab = p.bias_table + head;
    }

    // tile t = SUBS x 8 KiB of K + SUBS x 8 KiB of V; wave w moves the 1 KiB chunks 4 i + w of each half
    const unsigned char *kg = p.kp + unit0 * 8192 + wave * 1024 + lane * 16;
    const unsigned char *vg = p.vp + unit0 * 8192 + wave * 1024 + lane * 16;
    auto dma = [&](int t, int buf) {
        unsigned char *s = smem_h2 + buf * TILE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2 * SUBS; ++i) {
            __builtin_amdgcn_global_load_lds((glob_void_t *)(kg + (int64_t)t * (SUBS * 8192) + i * 4096),
                                             (lds_void_t *)(s + i * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glob_void_t *)(vg + (int64_t)t * (SUBS * 8192) + i * 4096),
                                             (lds_void_t *)(s + SUBS * 8192 + i * 4096), 16, 0, 0);
        }
    };

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float cs = HAS_BIAS ? 1.44269504088896340736f : p.s_unscale * 1.44269504088896340736f;

    const int ntiles = p.N / (32 * SUBS);
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if ((DBG & 16) == 0 && t + 1 < ntiles) dma(t + 1, buf ^ 1);
        const unsigned char *Ks = smem_h2 + buf * TILE + hi * 512 + r32 * 16;
        const unsigned char *Vs = Ks + SUBS * 8192;

        // fragment batches: 8 x ds_read_b128 each
        auto load_k = [&](int sub, u32x4 (&f)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const u32x4 *>(Ks + sub * 8192 + i * 1024);  // [plane][ks]
        };
        auto load_v = [&](int sub, u32x4 (&f)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const u32x4 *>(Vs + sub * 8192 + i * 1024);  // [plane][j * 2 + mt]
        };
        auto qk = [&](int sub, const u32x4 (&kf)[8]) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
            if constexpr (DBG & 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = __builtin_bit_cast(float, kf[r & 7][r >> 3] ^ qf[0][r & 3][r >> 2]) * 1e-30f;
                return st;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kh = __builtin_bit_cast(f16x8, kf[ks]), kl = __builtin_bit_cast(f16x8, kf[4 + ks]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, __builtin_bit_cast(f16x8, qf[0][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[1][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[0][ks]), st, 0, 0, 0);
            }
            if constexpr (HAS_BIAS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * (32 * SUBS) + sub * 32 + mfma32_row(r, hi);
                    const int ky = kv / p.gw, kx = kv % p.gw;
                    st[r] = fmaf(st[r], p.s_unscale,
                                 btab[((qy - ky + p.gh - 1) * (2 * p.gw - 1) + (qx - kx + p.gw - 1)) * p.heads]);
                }
            }
            return st;
        };
        auto softmax = [&](f32x16 &st, f16x8 (&ph)[2], f16x8 (&pl)[2]) {
            float alpha = 1.0f;
            if constexpr ((DBG & 1) == 0) {
                float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
                mx = halves_max(fmaxf(mx, st[15]));
                const float m_new = fmaxf(m_run, mx);
                const float mc = fmaf(m_new, cs, -P_SHIFT);
                float ps = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], cs, -mc));
                    ps += st[r];
                }
                ps = halves_sum(ps);
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
                l_run = fmaf(l_run, alpha, ps);
                m_run = m_new;
            } else {
                l_run += st[0];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pa[e] = st[8 * j + e];
                    pb[e] = st[8 * j + 4 + e];
                }
                const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                ph[j] = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (DBG & 32) {
                    pl[j] = ph[j];
                } else {
                    const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                    const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                    pl[j] = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            return alpha;
        };
        auto rescale = [&](float alpha) {
            if constexpr (DBG & 1) return;
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            }
        };
        auto pv = [&](const u32x4 (&vf)[8], const f16x8 (&ph)[2], const f16x8 (&pl)[2]) {
            if constexpr (DBG & 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ot[0][r] += __builtin_bit_cast(float, vf[r & 7][r >> 3]) * (float)ph[0][r & 7];
                    ot[1][r] += __builtin_bit_cast(float, vf[(r + 1) & 7][r >> 3]) * (float)pl[1][r & 7];
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, vf[j * 2 + mt]), vl = __builtin_bit_cast(f16x8, vf[4 + j * 2 + mt]);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[j], ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[j], ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[j], ot[mt], 0, 0, 0);
                }
        };

        if constexpr (SUBS == 2) {
            u32x4 ka[8], kb[8], va[8], vb[8];
            f16x8 pha[2], pla[2], phb[2], plb[2];
            load_k(0, ka);
            load_k(1, kb);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sa = qk(0, ka);
            __builtin_amdgcn_sched_barrier(0);
            load_v(0, va);                            // lands under the S^T MFMAs of b and the softmax of a
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sb = qk(1, kb);
            __builtin_amdgcn_sched_barrier(0);
            const float alpha_a = softmax(sa, pha, pla);
            rescale(alpha_a);
            load_v(1, vb);                            // lands under the P.V MFMAs of a and the softmax of b
            __builtin_amdgcn_sched_barrier(0);
            pv(va, pha, pla);
            __builtin_amdgcn_sched_barrier(0);
            const float alpha_b = softmax(sb, phb, plb);
            rescale(alpha_b);
            pv(vb, phb, plb);
        } else {
            u32x4 ka[8], va[8];
            f16x8 pha[2], pla[2];
            load_k(0, ka);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sa = qk(0, ka);
            __builtin_amdgcn_sched_barrier(0);
            load_v(0, va);
            __builtin_amdgcn_sched_barrier(0);
            const float alpha_a = softmax(sa, pha, pla);
            rescale(alpha_a);
            pv(va, pha, pla);
        }
        if constexpr ((DBG & 8) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    if (!wave_active) return;
    float bound = p.v_bound;
    if (p.v_bound_dev) bound *= p.v_bound_dev[(int64_t)p.v_bound_stride * (seq / p.seq_per_clip)];
    const float inv_l = 1.0f / (l_run * h2_scale_of_bound(bound));
    if (p.out_planes) {
        const float so = h2_scale_of_bound(bound);
        const int64_t row = (int64_t)seq * p.N + q_local;
        if (head == 0 && hi == 0) p.out_scale[row] = 1.0f / so;
#pragma unroll
        for (int d = 0; d < 2; ++d) pl_store_ot_block(ot[d], inv_l * so, p.out_planes, row, head * 2 + d, p.heads * 2, hi);
        return;
    }
    float *orow = p.out + ((int64_t)seq * p.N + q_local) * p.ldo + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[d][g * 4 + e] * inv_l;
            *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
        }
}


// -------------------------------------------------------------------------------------------
// Variant 4 ("attn_h2_variant" 4): 64 queries per wave.  The ablation arms of variant 3 (profiles/r04_attn_ablation.txt)
// price the launch at  MFMA 0.53 ms + softmax VALU 0.22 ms + LDS fragment reads / DMA / barriers 0.41 ms, almost additive
// (1.00 ms measured): a wave reads the whole 16 KiB K|V sub-block from LDS for 24 MFMAs, and an in-order wave runs its
// phases one after the other.  Here a wave owns TWO 32-query blocks (A, B) of the same (sequence, head):
//   * every K / V fragment read from LDS feeds 6 MFMAs instead of 3 (half the LDS bytes per flop, half the barriers and
//     DMA issues per flop: a workgroup covers 256 queries);
//   * the two blocks are independent, so the softmax of A is interleaved INTO the S^T MFMAs of B and the softmax of B into
//     the P.V MFMAs of A by sched_group_barrier pipelines (1 MFMA : ~10 VALU) -- matrix and vector work of ONE wave overlap.
// Arithmetic per query block is that of variants 1 - 3, bit for bit.  ILV = false: the same kernel without the pipelines.
// -------------------------------------------------------------------------------------------
//
// OPT (variant 6): two cuts of the softmax's VALU work, which the ablation shows is paid in full beside the MFMAs --
//   * the rescale of O is DEFERRED (cdna guide T13): a block's row maximum replaces the reference maximum only when it
//     exceeds it by more than 2^6.5 (wave-uniform decision, taken before the block's P is formed, so nothing is scaled
//     twice or not at all); otherwise P = exp(s - m_ref) may reach 2^6.5, which is why P is scaled by 2^9 instead of 2^14
//     before the fp16 split (2^15.5 < 65504; the split stays a 22-bit hi|lo pair, and O / l is unchanged in exact
//     arithmetic).  On l2-normalised q, k the running maximum settles within the first blocks: the 32 multiplies per
//     block and query block, taken ~85 % of the time before, become rare;
//   * the lo plane of P comes from v_fma_mix_f32 (hi as fp16 operand: lo = p - hi in one instruction instead of a
//     conversion and a subtraction): 4 instead of 6 instructions per pair of probabilities.
// NWAVE = 8 (variant 7): 512 queries per workgroup, one workgroup per CU -- half the L2 -> LDS traffic and barriers per flop again.
template <bool HAS_BIAS, bool ILV, bool OPT = false, int NWAVE = 4>
__global__ __launch_bounds__(64 * NWAVE, NWAVE == 4 ? 2 : 1) void attn_spatial_h2w_kernel(AttnH2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    constexpr int TILE = 16384;  // K block (8 KiB) | V block (8 KiB) of 32 keys
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nqb = p.nqb;       // 256-query blocks per (sequence, head)
    const int b = blockIdx.x, jb = b >> 3;
    const int grp = (jb / nqb) * 8 + (b & 7);
    if (grp >= p.ngrp_real) return;
    const int qb = jb % nqb, head = grp % p.heads, seq = grp / p.heads;
    const int nblk = p.N / 32;
    const int64_t unit0 = ((int64_t)seq * p.heads + head) * nblk;
    bool active[2];
    int q_local[2];
    u32x4 qf[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int qblk_raw = qb * (2 * NWAVE) + wave * 2 + i;
        active[i] = qblk_raw < nblk;
        const int qblk = active[i] ? qblk_raw : 0;
        q_local[i] = qblk * 32 + r32;
        const unsigned char *qb_ = p.qp + (unit0 + qblk) * 8192 + hi * 512 + r32 * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[i][pl][ks] = *reinterpret_cast<const u32x4 *>(qb_ + (pl * 4 + ks) * 1024);
    }
    // Legacy relative-position bias (reference attention.py:453-483, ContinuousPositionBias :535-583): bias[dy][dx] of THIS head,
    // (2 gh - 1) x (2 gw - 1) floats, staged ONCE per workgroup in LDS behind the K / V ring (32 x 32 grid: 15.9 KiB).  A score's
    // bias address is then   table + qpos(lane) - base(t) - c(r)   bytes with
    //   qpos = the query's (qy + gh - 1, qx + gw - 1) position minus this lane half's 4 key columns   (per lane and query block)
    //   base = position of the tile's first key (ky, kx)                                             (wave-uniform, per key tile)
    //   c    = position of accumulator register r's key row inside the tile                           (wave-uniform, 16 scalars)
    // -- one v_sub and one ds_read_b32 per score.  r02-r05 gathered the table from global memory with a division per score and
    // 64-bit addresses: the 64-query kernels spilled on it and the bias configurations stayed on the 32-query variant 3.
    int qpos[2] = {0, 0};
    int c_r[16];
    const unsigned char *tabp = smem_h2 + 2 * TILE;
    if constexpr (HAS_BIAS) {
        const int pitch = 2 * p.gw - 1, P = (2 * p.gh - 1) * pitch;
        float *tab = reinterpret_cast<float *>(smem_h2 + 2 * TILE);
        for (int idx = tid; idx < P; idx += 64 * NWAVE) tab[idx] = p.bias_table[(int64_t)idx * p.heads + head];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qy = q_local[i] >> p.lg_gw, qx = q_local[i] & (p.gw - 1);
            qpos[i] = ((qy + p.gh - 1) * pitch + qx + p.gw - 1 - 4 * hi) * 4;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row8 = 8 * (r >> 2);   // key row of register r inside the 32-key tile, without the lane half's + 4 hi
            c_r[r] = __builtin_amdgcn_readfirstlane(((row8 >> p.lg_gw) * pitch + (row8 & (p.gw - 1)) + (r & 3)) * 4);
        }
    }

    const unsigned char *kg = p.kp + unit0 * 8192 + wave * 1024 + lane * 16;
    const unsigned char *vg = p.vp + unit0 * 8192 + wave * 1024 + lane * 16;
    auto dma = [&](int t, int buf) {  // 8 pieces of 1 KiB per matrix: piece NWAVE i + wave
        unsigned char *s = smem_h2 + buf * TILE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 8 / NWAVE; ++i) {
            __builtin_amdgcn_global_load_lds((glob_void_t *)(kg + (int64_t)t * 8192 + i * (NWAVE * 1024)),
                                             (lds_void_t *)(s + i * (NWAVE * 1024)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glob_void_t *)(vg + (int64_t)t * 8192 + i * (NWAVE * 1024)),
                                             (lds_void_t *)(s + 8192 + i * (NWAVE * 1024)), 16, 0, 0);
        }
    };

    f32x16 ot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[i][d][r] = 0.0f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};
    const float cs = HAS_BIAS ? 1.44269504088896340736f : p.s_unscale * 1.44269504088896340736f;

    const int ntiles = p.N / 32;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) dma(t + 1, buf ^ 1);
        const unsigned char *Ks = smem_h2 + buf * TILE + hi * 512 + r32 * 16;
        const unsigned char *Vs = Ks + 8192;
        u32x4 kf[8], vf[8];
        auto qk = [&](int i) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kh = __builtin_bit_cast(f16x8, kf[ks]), kl = __builtin_bit_cast(f16x8, kf[4 + ks]);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, __builtin_bit_cast(f16x8, qf[i][0][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[i][1][ks]), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[i][0][ks]), st, 0, 0, 0);
            }
            if constexpr (HAS_BIAS) {
                const int k0 = t * 32;
                const int base_t = __builtin_amdgcn_readfirstlane(((k0 >> p.lg_gw) * (2 * p.gw - 1) + (k0 & (p.gw - 1))) * 4);
                const unsigned char *tq = tabp + (qpos[i] - base_t);
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = fmaf(st[r], p.s_unscale, *reinterpret_cast<const float *>(tq - c_r[r]));
            }
            return st;
        };
        auto rescale = [&](int i, float alpha) {
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[i][d][r] *= alpha;
            }
        };
        auto softmax = [&](int i, f32x16 &st, f16x8 (&ph)[2], f16x8 (&pl)[2]) {
            float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
            mx = halves_max(fmaxf(mx, st[15]));
            float alpha = 1.0f, mc;
            if constexpr (OPT) {
                // m_run holds the reference maximum in log2 units
                const float mxs = mx * cs;
                if (__any(mxs > m_run[i] + 6.5f)) {
                    const float m_new = fmaxf(m_run[i], mxs);
                    const float a = __builtin_amdgcn_exp2f(m_run[i] - m_new);  // 0 on the first block (m_run = -inf)
                    l_run[i] *= a;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ot[i][d][r] *= a;
                    m_run[i] = m_new;
                }
                mc = m_run[i] - 9.0f;
            } else {
                const float m_new = fmaxf(m_run[i], mx);
                mc = fmaf(m_new, cs, -P_SHIFT);
                alpha = __builtin_amdgcn_exp2f((m_run[i] - m_new) * cs);
                m_run[i] = m_new;
            }
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], cs, -mc));
                ps += st[r];
            }
            ps = halves_sum(ps);
            if constexpr (OPT) l_run[i] += ps;
            else l_run[i] = fmaf(l_run[i], alpha, ps);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (OPT) {
                    u32x4 h4, l4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = st[8 * j + 2 * e], b2 = st[8 * j + 2 * e + 1];
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b2}, f16x2_t));
                        float ra, rb;
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hp), "v"(a));
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hp), "v"(b2));
                        h4[e] = hp;
                        l4[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{ra, rb}, f16x2_t));
                    }
                    ph[j] = __builtin_bit_cast(f16x8, h4);
                    pl[j] = __builtin_bit_cast(f16x8, l4);
                } else {
                    f32x4 pa, pb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pa[e] = st[8 * j + e];
                        pb[e] = st[8 * j + 4 + e];
                    }
                    const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                    const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                    const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                    ph[j] = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                    pl[j] = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            return alpha;
        };
        auto pv = [&](int i, const f16x8 (&ph)[2], const f16x8 (&pl)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, vf[j * 2 + mt]), vl = __builtin_bit_cast(f16x8, vf[4 + j * 2 + mt]);
                    ot[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[j], ot[i][mt], 0, 0, 0);
                    ot[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[j], ot[i][mt], 0, 0, 0);
                    ot[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[j], ot[i][mt], 0, 0, 0);
                }
        };
        // 12 x (1 MFMA : 10 VALU / transcendental) for the region that ends here
        auto pipeline = [&]() {
            if constexpr (ILV) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, 10, 0);
                }
            }
        };

        f16x8 pha[2], pla[2], phb[2], plb[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = *reinterpret_cast<const u32x4 *>(Ks + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sa = qk(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) vf[i] = *reinterpret_cast<const u32x4 *>(Vs + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sb = qk(1);                                  // S^T(B) MFMAs ...
        const float alpha_a = softmax(0, sa, pha, pla);     // ... with the softmax of A in their shadow
        pipeline();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!OPT) rescale(0, alpha_a);
        pv(0, pha, pla);                                    // P.V(A) MFMAs ...
        const float alpha_b = softmax(1, sb, phb, plb);     // ... with the softmax of B in their shadow
        pipeline();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!OPT) rescale(1, alpha_b);
        pv(1, phb, plb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float bound = p.v_bound;
    if (p.v_bound_dev) bound *= p.v_bound_dev[(int64_t)p.v_bound_stride * (seq / p.seq_per_clip)];
    const float so = h2_scale_of_bound(bound);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!active[i]) continue;
        const float inv_l = 1.0f / (l_run[i] * so);
        const int64_t row = (int64_t)seq * p.N + q_local[i];
        if (p.out_planes) {
            if (head == 0 && hi == 0) p.out_scale[row] = 1.0f / so;
#pragma unroll
            for (int d = 0; d < 2; ++d) pl_store_ot_block(ot[i][d], inv_l * so, p.out_planes, row, head * 2 + d, p.heads * 2, hi);
            continue;
        }
        float *orow = p.out + row * p.ldo + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ot[i][d][g * 4 + e] * inv_l;
                *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
            }
    }
}


// -------------------------------------------------------------------------------------------
// Window attention on packed operands (reference attention.py:254-293): one wave per (frame, window, head) -- 64 queries
// x 64 keys x 64 channels -- with Q, K, V as the 8 KiB fragment-order blocks the q|k|v plane GEMM's packing epilogues
// write when its rows arrive in window-major order (omnitok_stats_pack_windows; n_tokens = 64, so a "sequence" of the
// packed layout is one window).  Every operand byte is read once, by one wave, as lane-linear 16-byte loads straight
// into MFMA fragments: no LDS, no staging, 48 KiB in and 16 KiB (planes) out per wave -- an HBM-bound kernel, which is
// what SURVEY 8(d) prices window attention as.  Arithmetic: the fp16 hi|lo split products of the spatial kernel,
// logits = q.k (q pre-multiplied by the head scale in the packing epilogue) + relative_position_bias[head][key][query]
// (dense [heads][64][64], built once at finalize), softmax over the window's 64 keys in fp32, P.V.
// The output rows go back to token order (frame, y, x): as fp16 hi|lo planes for the proj GEMM, or fp32 rows.
// -------------------------------------------------------------------------------------------
struct WinH2Params {
    const unsigned char *qp, *kp, *vp;
    const float *bias_dense;  // [heads][64 kv][64 q]
    float *out; int64_t ldo;
    unsigned char *out_planes;
    float out_mul;            // power-of-two scale of the output planes (static bound of |V|)
    float s_unscale, v_scale;  // 1 / (sq sk); the V planes' power-of-two scale
    int gh, gw, heads, nwin_x, nwin;
};

__global__ __launch_bounds__(256, 2) void attn_window_h2_kernel(WinH2Params p, int64_t total_units) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (frame, window, head), head fastest
    if (unit >= total_units) return;
    const int r32 = lane & 31, hi = lane >> 5;
    const int head = (int)(unit % p.heads);
    const int64_t seq = unit / p.heads;           // (frame, window)
    const int win = (int)(seq % p.nwin);
    const int64_t frame = seq / p.nwin;
    const int wy = win / p.nwin_x, wx = win % p.nwin_x;
    const int64_t base = (seq * p.heads + head) * 2 * 8192 + hi * 512 + r32 * 16;

    // K fragments of both 32-key blocks: [kb][plane * 4 + ks]
    u32x4 kf[2][8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[kb][i] = *reinterpret_cast<const u32x4 *>(p.kp + base + kb * 8192 + i * 1024);
    const float *bh = p.bias_dense + (int64_t)head * 64 * 64;
    const float LOG2E = 1.44269504088896340736f;
    // V fragments of both key blocks: requested ONCE, before the S^T MFMAs of the first query block (they land under them and
    // the softmax) and kept for the second -- re-reading them per query block showed as 1.33x the operand bytes on the
    // memory-side counter (profiles/r04_pmc_FETCH_SIZE.csv)
    u32x4 vf[2][8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i) vf[kb][i] = *reinterpret_cast<const u32x4 *>(p.vp + base + kb * 8192 + i * 1024);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        u32x4 qf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const u32x4 *>(p.qp + base + qb * 8192 + i * 1024);
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kh = __builtin_bit_cast(f16x8, kf[kb][ks]), kl = __builtin_bit_cast(f16x8, kf[kb][4 + ks]);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, __builtin_bit_cast(f16x8, qf[ks]), st[kb], 0, 0, 0);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[4 + ks]), st[kb], 0, 0, 0);
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, __builtin_bit_cast(f16x8, qf[ks]), st[kb], 0, 0, 0);
            }
        }
        // logits (attention.py:274-282) and the softmax over the window's 64 keys (both blocks are in registers: one pass)
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kb * 32 + mfma32_row(r, hi);
                const float sgt = fmaf(st[kb][r], p.s_unscale, bh[kv * 64 + qb * 32 + r32]);
                st[kb][r] = sgt;
                mx = fmaxf(mx, sgt);
            }
        mx = halves_max(mx);
        const float mc = fmaf(mx, LOG2E, -P_SHIFT);
        float ps = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[kb][r] = __builtin_amdgcn_exp2f(fmaf(st[kb][r], LOG2E, -mc));
                ps += st[kb][r];
            }
        ps = halves_sum(ps);
        f32x16 ot[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pa[e] = st[kb][8 * j + e];
                    pb[e] = st[kb][8 * j + 4 + e];
                }
                const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                const f16x8 ph = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
                const f16x8 pl = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, vf[kb][j * 2 + mt]), vl = __builtin_bit_cast(f16x8, vf[kb][4 + j * 2 + mt]);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, ot[mt], 0, 0, 0);
                    ot[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, ot[mt], 0, 0, 0);
                }
            }
        // token row of this lane's query (window position qb * 32 + r32), reference window_reverse (attention.py:191-207)
        const int pos = qb * 32 + r32;
        const int64_t row = frame * (int64_t)p.gh * p.gw + (int64_t)(wy * 8 + (pos >> 3)) * p.gw + wx * 8 + (pos & 7);
        const float inv_l = 1.0f / (ps * p.v_scale);   // ps carries 2^14, the V planes their power-of-two scale: both exact
        if (p.out_planes) {
#pragma unroll
            for (int d = 0; d < 2; ++d) pl_store_ot_block(ot[d], inv_l * p.out_mul, p.out_planes, row, head * 2 + d, p.heads * 2, hi);
        } else {
            float *orow = p.out + row * p.ldo + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ot[d][g * 4 + e] * inv_l;
                    *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
                }
        }
    }
}

int g_attn_h2_dbg = 0;  // "attn_h2_dbg": measurement builds of variant 3 (OMNITOK_ATTN_MEASUREMENT_BUILDS only)
// "attn_h2_variant": 6 (default) 64 queries per wave, deferred rescale, v_fma_mix split | 5 64 queries per wave, softmax of
// variants 1 - 3 | 4 = 5 with sched_group_barrier pipelines | 3 32 queries per wave, hoisted fragment reads, 32-key tiles, three
// workgroups per CU (also what the legacy-bias path runs) | 2 the same with 64-key tiles | 1 the r02 / r03 pipelined + LDS-DMA kernel
// | 0 the first kernel.  One box, C3 shape (profiles/r04_attn_variants.txt): 1.109 / 1.007 / 0.929 / 0.916 ms for 1 / 3 / 5 / 6.
int g_attn_h2_variant = 6;
//  // "attn_h2_variant": 1 pipelined + LDS-DMA kernel (default), 0 the first kernel

}  // namespace omnitok

using namespace omnitok;

// q_bound / k_bound: upper bounds of |q| (after RoPE, l2norm, q_scale and `scale`) and |k|: scale * max|q_scale|
// and max|k_scale| (l2-normalised vectors have |x_d| <= 1).  v_bound (x v_bound_dev[stride * clip] when given):
// upper bound of |v| for the rows of a clip (rows_per_clip consecutive rows).
extern "C" int omnitok_attn_pack(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv,
                                 int64_t rows, int n_tokens, int heads, const float *cos, const float *sin,
                                 const float *q_scale, const float *k_scale, float scale, float q_bound, float k_bound,
                                 float v_bound, const float *v_bound_dev, int v_bound_stride, int64_t rows_per_clip,
                                 void *qp, void *kp, void *vp, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(q && k && q_scale && k_scale && qp && kp, "attn_pack: null pointer");
    OT_CHECK_ARG((v == nullptr) == (vp == nullptr), "attn_pack: v and vp must both be given or both null (V packed elsewhere)");
    OT_CHECK_ARG((cos == nullptr) == (sin == nullptr), "attn_pack: cos/sin must both be given or both null");
    OT_CHECK_ARG(ldq % 4 == 0 && ldkv % 4 == 0 && aligned16(q) && aligned16(k) && (!v || aligned16(v)) && aligned16(qp) &&
                     aligned16(kp) && (!vp || aligned16(vp)), "attn_pack: unaligned");
    OT_CHECK_ARG(n_tokens > 0 && n_tokens % 32 == 0 && rows % n_tokens == 0, "attn_pack: %lld rows of %d-token sequences",
                 (long long)rows, n_tokens);
    OT_CHECK_ARG(q_bound > 0.0f && k_bound > 0.0f && (!v || v_bound > 0.0f), "attn_pack: operand bounds must be positive");
    OT_CHECK_ARG(!v_bound_dev || (rows_per_clip > 0 && rows_per_clip % n_tokens == 0),
                 "attn_pack: rows_per_clip must be a whole number of sequences");
    if (rows == 0) return OMNITOK_OK;
    PackParams p;
    p.q = q; p.ldq = ldq; p.k = k; p.v = v; p.ldkv = ldkv; p.n_tokens = n_tokens; p.heads = heads;
    p.nblk = n_tokens / 32; p.cosT = cos; p.sinT = sin; p.q_scale = q_scale; p.k_scale = k_scale; p.scale = scale;
    p.sq = h2_scale_of_bound(q_bound); p.sk = h2_scale_of_bound(k_bound);
    p.v_bound = v_bound; p.v_bound_dev = v_bound_dev; p.v_bound_stride = v_bound_stride;
    p.rows_per_clip = v_bound_dev ? rows_per_clip : rows;
    p.qp = static_cast<unsigned char *>(qp); p.kp = static_cast<unsigned char *>(kp);
    p.vp = static_cast<unsigned char *>(vp);
    const int64_t units = rows / 32 * heads;
    OT_CHECK_ARG(units < (1ll << 31), "attn_pack: grid too large");
    hipLaunchKernelGGL(attn_pack_kernel, dim3((unsigned)units), dim3(256), 0, stream, p);
    OT_LAUNCH_CHECK("attn_pack");
    return OMNITOK_OK;
}

extern "C" int omnitok_attn_spatial_h2(const void *qp, const void *kp, const void *vp, float *out, int64_t ldo, int Bn,
                                       int N, int heads, float q_bound, float k_bound, float v_bound,
                                       const float *v_bound_dev, int v_bound_stride, int seq_per_clip,
                                       const float *bias_table, int gh, int gw, omnitok_stream_t stream_) {
    return omnitok_attn_spatial_h2_planes(qp, kp, vp, out, ldo, nullptr, nullptr, Bn, N, heads, q_bound, k_bound, v_bound,
                                          v_bound_dev, v_bound_stride, seq_per_clip, bias_table, gh, gw, stream_);
}

extern "C" int omnitok_attn_spatial_h2_planes(const void *qp, const void *kp, const void *vp, float *out, int64_t ldo,
                                              void *out_planes, float *out_scale, int Bn, int N, int heads, float q_bound,
                                              float k_bound, float v_bound, const float *v_bound_dev, int v_bound_stride,
                                              int seq_per_clip, const float *bias_table, int gh, int gw,
                                              omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(qp && kp && vp && (out || out_planes), "attn_spatial_h2: null pointer");
    OT_CHECK_ARG(!out_planes || (out_scale && aligned16(out_planes)), "attn_spatial_h2: plane output needs out_scale");
    OT_CHECK_ARG(N % 64 == 0 && N > 0, "attn_spatial_h2: N=%d must be a multiple of 64 tokens", N);
    OT_CHECK_ARG(!bias_table || gh * gw == N, "attn_spatial_h2: bias grid %dx%d != N=%d", gh, gw, N);
    OT_CHECK_ARG((out_planes || ldo % 4 == 0) && aligned16(qp) && aligned16(kp) && aligned16(vp) && (!out || aligned16(out)),
                 "attn_spatial_h2: unaligned");
    OT_CHECK_ARG(q_bound > 0.0f && k_bound > 0.0f && v_bound > 0.0f, "attn_spatial_h2: operand bounds must be positive");
    OT_CHECK_ARG(!v_bound_dev || seq_per_clip > 0, "attn_spatial_h2: seq_per_clip");
    OT_CHECK_ARG((int64_t)heads * Bn * ((N + 127) / 128) < (1ll << 31) - 8, "attn_spatial_h2: grid too large");
    if (Bn == 0) return OMNITOK_OK;
    AttnH2Params p;
    p.qp = static_cast<const unsigned char *>(qp); p.kp = static_cast<const unsigned char *>(kp);
    p.vp = static_cast<const unsigned char *>(vp);
    p.out = out; p.ldo = ldo; p.N = N; p.heads = heads;
    p.s_unscale = 1.0f / (h2_scale_of_bound(q_bound) * h2_scale_of_bound(k_bound));
    p.v_bound = v_bound; p.v_bound_dev = v_bound_dev; p.v_bound_stride = v_bound_stride;
    p.seq_per_clip = v_bound_dev ? seq_per_clip : (Bn > 0 ? Bn : 1);
    p.bias_table = bias_table; p.gh = gh; p.gw = gw;
    p.out_planes = static_cast<unsigned char *>(out_planes); p.out_scale = out_scale;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2_kernel<false>), AH_LDS_BYTES)) return rc;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2_kernel<true>), AH_LDS_BYTES)) return rc;
    p.nqb = (N + 127) / 128;
    const int ngrp = ((heads * Bn + 7) / 8) * 8;
    p.ngrp_real = heads * Bn;
    dim3 grid((unsigned)((int64_t)ngrp * p.nqb));
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2p_kernel<false>), AH_LDS_BYTES)) return rc;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2p_kernel<true>), AH_LDS_BYTES)) return rc;
    // legacy bias on the 64-query kernels (variants 4 - 7): the head's table lives in LDS (see attn_spatial_h2w_kernel), which needs
    // a power-of-two grid width >= 8 (position by shift / mask) and a table of at most 32 KiB (grids up to 32 x 64: two workgroups
    // per CU keep fitting); anything else -- and variant 7's 512-query form, not built with a bias -- takes variant 3's global gather
    int lg = -1;
    if (bias_table && gw >= 8 && (gw & (gw - 1)) == 0 && (int64_t)(2 * gh - 1) * (2 * gw - 1) * 4 <= 32768) {
        lg = 0;
        while ((1 << lg) < gw) ++lg;
    }
    p.lg_gw = lg;
    const int bias_lds = lg >= 0 ? (((2 * gh - 1) * (2 * gw - 1) * 4 + 15) & ~15) : 0;
    int variant = g_attn_h2_variant;
    if (bias_table && variant >= 4) variant = lg >= 0 ? (variant == 7 ? 6 : variant) : 3;
    if (variant == 7 && N % 512 == 0) {
        p.nqb = N / 512;
        dim3 gridw((unsigned)((int64_t)ngrp * p.nqb));
        hipLaunchKernelGGL((attn_spatial_h2w_kernel<false, false, true, 8>), gridw, dim3(512), 2 * 16384, stream, p);
    } else if (variant == 6 || variant == 7) {
        p.nqb = (N + 255) / 256;
        dim3 gridw((unsigned)((int64_t)ngrp * p.nqb));
        if (bias_table) {
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2w_kernel<true, false, true>), 2 * 16384 + bias_lds)) return rc;
            hipLaunchKernelGGL((attn_spatial_h2w_kernel<true, false, true>), gridw, dim3(256), 2 * 16384 + bias_lds, stream, p);
        } else hipLaunchKernelGGL((attn_spatial_h2w_kernel<false, false, true>), gridw, dim3(256), 2 * 16384, stream, p);
    } else if (variant == 4 || variant == 5) {
        // 256 queries per workgroup
        p.nqb = (N + 255) / 256;
        dim3 gridw((unsigned)((int64_t)ngrp * p.nqb));
        const int lds = 2 * 16384;
        const bool ilv = variant == 4;
        if (bias_table) {
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2w_kernel<true, true>), lds + bias_lds)) return rc;
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2w_kernel<true, false>), lds + bias_lds)) return rc;
            if (ilv) hipLaunchKernelGGL((attn_spatial_h2w_kernel<true, true>), gridw, dim3(256), lds + bias_lds, stream, p);
            else hipLaunchKernelGGL((attn_spatial_h2w_kernel<true, false>), gridw, dim3(256), lds + bias_lds, stream, p);
        } else {
            if (ilv) hipLaunchKernelGGL((attn_spatial_h2w_kernel<false, true>), gridw, dim3(256), lds, stream, p);
            else hipLaunchKernelGGL((attn_spatial_h2w_kernel<false, false>), gridw, dim3(256), lds, stream, p);
        }
    } else if (variant == 2 || variant == 3) {
        const bool v3 = variant == 3;
        const int lds = v3 ? 2 * 16384 : 2 * 32768;
        if (v3) {
#ifdef OMNITOK_ATTN_MEASUREMENT_BUILDS  // wrong-result ablation arms (profiles/r04_attn_ablation.txt)
#define OT_ATTN_DBG(D) case D: hipLaunchKernelGGL((attn_spatial_h2x_kernel<false, 1, D>), grid, dim3(256), lds, stream, p); break;
            if (g_attn_h2_dbg > 0 && !bias_table) {
                switch (g_attn_h2_dbg) {
                    OT_ATTN_DBG(1) OT_ATTN_DBG(2) OT_ATTN_DBG(4) OT_ATTN_DBG(6) OT_ATTN_DBG(7) OT_ATTN_DBG(8) OT_ATTN_DBG(24) OT_ATTN_DBG(32)
                    OT_ATTN_DBG(33)
                    case 100: hipLaunchKernelGGL((attn_spatial_h2x_kernel<false, 1, 0, 4>), grid, dim3(256), lds, stream, p); break;
                    case 101: hipLaunchKernelGGL((attn_spatial_h2x_kernel<false, 1, 0, 2>), grid, dim3(256), lds, stream, p); break;
                    case 102: hipLaunchKernelGGL((attn_spatial_h2x_kernel<false, 1, 0, 1>), grid, dim3(256), lds, stream, p); break;
                    default: set_error("attn_h2_dbg %d not built", g_attn_h2_dbg); return OMNITOK_ERR_INVALID;
                }
                OT_LAUNCH_CHECK("attn_spatial_h2 (measurement build)");
                return OMNITOK_OK;
            }
#endif
            if (bias_table)
                hipLaunchKernelGGL((attn_spatial_h2x_kernel<true, 1>), grid, dim3(256), lds, stream, p);
            else
                hipLaunchKernelGGL((attn_spatial_h2x_kernel<false, 1>), grid, dim3(256), lds, stream, p);
        } else {
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2x_kernel<false, 2>), lds)) return rc;
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_h2x_kernel<true, 2>), lds)) return rc;
            if (bias_table)
                hipLaunchKernelGGL((attn_spatial_h2x_kernel<true, 2>), grid, dim3(256), lds, stream, p);
            else
                hipLaunchKernelGGL((attn_spatial_h2x_kernel<false, 2>), grid, dim3(256), lds, stream, p);
        }
    } else if (variant == 1) {
        if (bias_table)
            hipLaunchKernelGGL(attn_spatial_h2p_kernel<true>, grid, dim3(256), AH_LDS_BYTES, stream, p);
        else
            hipLaunchKernelGGL(attn_spatial_h2p_kernel<false>, grid, dim3(256), AH_LDS_BYTES, stream, p);
    } else if (bias_table)
        hipLaunchKernelGGL(attn_spatial_h2_kernel<true>, grid, dim3(256), AH_LDS_BYTES, stream, p);
    else
        hipLaunchKernelGGL(attn_spatial_h2_kernel<false>, grid, dim3(256), AH_LDS_BYTES, stream, p);
    OT_LAUNCH_CHECK("attn_spatial_h2");
    return OMNITOK_OK;
}

// Window attention over packed operands (see attn_window_h2_kernel).  qp / kp / vp: the packed planes of Bn * (gh / 8) *
// (gw / 8) windows x heads, 64 tokens each, rows in window-major order; q_bound / k_bound / v_bound: the static bounds the
// packing epilogues scaled them with.  out (fp32 [Bn * gh * gw, ldo], token order) or out_planes (hi|lo planes, K = heads *
// 64, scaled by the power of two of v_bound: the consumer's a_scale_const = omnitok_pl_unscale(v_bound)).
extern "C" int omnitok_attn_window_h2(const void *qp, const void *kp, const void *vp, const float *bias_dense, float *out,
                                      int64_t ldo, void *out_planes, float q_bound, float k_bound, float v_bound, int Bn, int gh,
                                      int gw, int heads, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(qp && kp && vp && bias_dense && (out || out_planes), "attn_window_h2: null pointer");
    OT_CHECK_ARG(gh % 8 == 0 && gw % 8 == 0, "attn_window_h2: grid %dx%d not divisible by the 8x8 window", gh, gw);
    OT_CHECK_ARG(aligned16(qp) && aligned16(kp) && aligned16(vp) && (!out || (aligned16(out) && ldo % 4 == 0)) &&
                     (!out_planes || aligned16(out_planes)), "attn_window_h2: unaligned");
    OT_CHECK_ARG(q_bound > 0.0f && k_bound > 0.0f && v_bound > 0.0f, "attn_window_h2: operand bounds must be positive");
    WinH2Params p;
    p.qp = static_cast<const unsigned char *>(qp); p.kp = static_cast<const unsigned char *>(kp);
    p.vp = static_cast<const unsigned char *>(vp);
    p.bias_dense = bias_dense; p.out = out; p.ldo = ldo; p.out_planes = static_cast<unsigned char *>(out_planes);
    p.v_scale = h2_scale_of_bound(v_bound);
    p.out_mul = p.v_scale;
    p.s_unscale = 1.0f / (h2_scale_of_bound(q_bound) * h2_scale_of_bound(k_bound));
    p.gh = gh; p.gw = gw; p.heads = heads; p.nwin_x = gw / 8; p.nwin = (gh / 8) * (gw / 8);
    const int64_t total = (int64_t)Bn * p.nwin * heads;
    if (total == 0) return OMNITOK_OK;
    OT_CHECK_ARG((total + 3) / 4 < (1ll << 31), "attn_window_h2: grid too large");
    hipLaunchKernelGGL(attn_window_h2_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, p, total);
    OT_LAUNCH_CHECK("attn_window_h2");
    return OMNITOK_OK;
}
