// fp32 GEMM  c = a . w^T (+ epilogue) on the fp16 matrix cores from a 2-way split of each operand
// ("h2" GEMM): three v_mfma_f32_32x32x16_f16 per 32x32x16 block instead of the six bf16 ones of
// gemm_x3.hip (and the eight fp32-input ones of gemm.hip).
//
// Why: measured on the MI355X (profiles/r02_x3_ablation.txt) the bf16x3 kernel is POWER bound: with
// every load, store and VALU instruction removed the MFMA stream alone runs at 261-276 TF
// fp32-equivalent (the chip clocks down to ~1.5 GHz under a dense bf16 MFMA stream), so the only
// lever left is fewer matrix-pipe passes per useful flop.
//
// Arithmetic.  Each fp32 operand element, scaled by a power of two (exact), is written as
//     a' = hi + lo + r,   hi = fp16(a'),  lo = fp16(a' - hi),  |r| <= 2^-22 |a'|   (11 + 11 bits)
// and the kernel accumulates hi.lo' + hi.hi' + lo.hi' in fp32 (the dropped lo.lo' and r terms are
// below 2^-21 |a b|; a CPU emulation and the GPU tests put the result error at or below that of the
// fp32 fmaf chain, whose accumulated rounding dominates).  fp16 has 5 exponent bits, so the scales
// matter: a weight row is scaled at load time so that its largest element is in [2^13, 2^14); the A
// operand is scaled by ONE power of two per launch, derived from an UPPER BOUND of |a| that the
// caller supplies (a static bound for LayerNorm outputs, a device-side bound from the row-statistics
// pass for raw activations) so that |a'| <= 2^15 < 65504 can never overflow.  Elements more than
// 2^18 below the bound lose relative (not absolute) precision: their error is <= 2^-25 of the
// scaled unit, i.e. <= 2^-40 of the bound.  The scales are removed exactly in the epilogue.
//
// Structure = gemm_x3.hip (persistent 256x256 tiles, 8 waves, BK = 16, two LDS stages, one LDS-only
// barrier per K-step, operands split while staged) with two planes per operand; the weight planes are
// pre-split / pre-scaled once (omnitok_h2_pack_weight: [N][K/8][hi|lo][8] fp16, the same 4 B per
// element as the fp32 weight) so only the A operand costs VALU work in the loop.
#include "gemm_x_common.h"
#include "h2_common.h"

#include <type_traits>

namespace omnitok {


extern int g_gemm_gn;
int g_h2_tile = 0;  // "h2_tile": 0 auto, 1 256x256, 3 128x128, 4 64x64 / 64x128 (A/B measurements)

struct H2Params {
    GemmParams g;            // g.w is unused (the weight comes as packed planes)
    const void *wpl;         // block layout of omnitok_h2_pack_weight
    const float *wscale;     // [N]: 2^f_n, the factor that undoes the row scaling
    float a_bound;           // static upper bound of |a|, multiplied by a_bound_dev[stride * clip] when set
    const float *a_bound_dev;
    int a_bound_stride;
    int64_t a_rpc;           // rows per clip (<= 0: one bound for all rows); tiles never straddle clips
    const float *ln_stats;   // as gemm_x3
    const float *ln_gamma;
    const float *ln_beta;
    int ln_cols;
    float ln_bound;          // upper bound of |LN(a)|
    float *c2;               // output columns [split_col, N) go to c2 (row stride ldc2) instead of c
    int64_t ldc2;
    int split_col;           // 0: everything to c
    // V columns of a merged q|k|v launch straight into the fp16 hi|lo planes of the attention kernel (attn_h2.hip):
    // the 32x32 accumulator layout IS the packed V layout (a lane's registers 8 j .. 8 j + 7 are one 16-byte fragment
    // of 16-key step j), so these tiles are never written as fp32
    unsigned char *vp = nullptr;   // packed V ([rows * heads * 64] x 4 bytes), or null
    int v_col0 = 0;                // first V column of the merged output (a multiple of the tile width)
    int v_ntok = 0, v_heads = 0;   // tokens per sequence (% 32 == 0), heads
    float v_bound = 0.0f;          // upper bound of |v| (times v_bound_dev[stride * clip] when set)
    const float *v_bound_dev = nullptr;
    int v_bound_stride = 1;
    int64_t v_rpc = 0;             // rows per clip
};

template <int WGM_, int WGN_, int MI_, int NI_>
struct H2Cfg {
    static constexpr int WGM = WGM_, WGN = WGN_, MI = MI_, NI = NI_;
    static constexpr int NT = 64 * WGM * WGN;
    static constexpr int TM = WGM * MI * 32, TN = WGN * NI * 32;
    static constexpr int RPP = NT / 4;
    static constexpr int NA = TM / RPP, NB = TN / RPP, NP = NA + NB;
    static constexpr int SUBA = TM * 16 + 64;   // A: ds_write_b64, k halves 64 B apart mod 128
    static constexpr int SUBB = TN * 16 + 32;   // W: ds_write_b128, (plane, k half) 32 B apart mod 128
    static constexpr int PLA = 2 * SUBA, PLB = 2 * SUBB;
    static constexpr int STAGE = 2 * PLA + 2 * PLB;
    static constexpr int LN_TAB = 2 * 512 * 4;
    static_assert(TM % RPP == 0 && TN % RPP == 0, "loader passes");
};

// DBG (measurement builds only, wrong results): 1 skip the split arithmetic, 2 skip the LDS stores of the
// K loop, 4 skip the global loads of the K loop, 8 skip the per-step barrier
template <int FLAGS, typename C, bool LN, int DBG = 0>
__global__ __launch_bounds__(C::NT, 2) void gemm_h2_kernel(H2Params hp) {
    constexpr int MI = C::MI, NI = C::NI, NA = C::NA, NB = C::NB, NP = C::NP;
    constexpr int TM = C::TM, TN = C::TN, RPP = C::RPP, SBK = 16;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const GemmParams &p = hp.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nk = p.K / SBK;
    if ((int)blockIdx.x >= p.ntiles) return;
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nbm = (int)((p.M + TM - 1) / TM), nbn = (p.N + TN - 1) / TN;

    // scales of the A operand: one power of two per clip for raw a (from the clip's range), one for LN(a)
    auto raw_scale = [&](int64_t tile_row0) {
        float b = hp.a_bound;
        if (hp.a_bound_dev) {
            const int64_t clip = hp.a_rpc > 0 ? tile_row0 / hp.a_rpc : 0;
            b *= hp.a_bound_dev[clip * hp.a_bound_stride];
        }
        return h2_scale_of_bound(b);
    };
    const float s_ln = LN ? h2_scale_of_bound(hp.ln_bound) : 1.0f;
    float cur_sraw = 1.0f;  // of the tile the load pointers address

    float *ln_tab = reinterpret_cast<float *>(smem_c + 2 * C::STAGE);
    if constexpr (LN) {
        for (int k = tid; k < p.K; k += C::NT) {
            ln_tab[2 * k] = hp.ln_gamma[k];
            ln_tab[2 * k + 1] = hp.ln_beta ? hp.ln_beta[k] : 0.0f;
        }
    }

    // ---- loader ------------------------------------------------------------------------------
    // A: thread -> k quad (tid & 3) of rows (tid >> 2) + RPP * i (fp32, 16 B);
    // W: thread -> 16-B chunk (tid & 3) = (k half, plane) of rows (tid >> 2) + RPP * i of the packed planes
    const int lrow = tid >> 2, lq = tid & 3;
    const float *ap[NA];
    const char *wp[NB];
    const float *sp[NA];  // (mean, rstd) of the rows the load pointers address
    bool cur_ln = false;
    auto set_ptrs = [&](int i) {
        const int lid = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, p.ntiles);
        int64_t tbm;
        int tbn;
        tile_coords(lid, nbm, nbn, p.gn, tbm, tbn);
        const bool ln_on = LN && tbn * TN < hp.ln_cols;
        cur_ln = ln_on;
        cur_sraw = raw_scale(tbm * TM);
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            int64_t gr = tbm * TM + lrow + RPP * r;
            if (gr > p.M - 1) gr = p.M - 1;
            int64_t ar = gr;
            if (p.a_rpg > 0) ar = (gr / p.a_rpg) * p.a_stride + p.a_off + (gr % p.a_rpg);
            ap[r] = p.a + ar * p.lda + lq * 4;
            if constexpr (LN) sp[r] = hp.ln_stats + 2 * ar;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            int wr = tbn * TN + lrow + RPP * r;
            if (wr > p.N - 1) wr = p.N - 1;
            // row wr of the block layout; the K-step's chunk (k half lq >> 1, plane lq & 1) is added per load
            wp[r] = static_cast<const char *>(hp.wpl) + (int64_t)(wr >> 6) * (p.K >> 5) * 8192 + (wr & 63) * 16 +
                    (lq & 1) * 4096 + (lq >> 1) * 1024;
        }
    };
    f32x4 ga[2][NA];
    u32x4 gw[2][NB];
    float st_mean[2][NA], st_rstd[2][NA];
    int st_tab[2];
    bool st_ln[2];
    float st_sraw[2];
    auto gload = [&](int set, int k0, bool in_loop = true) {
        if constexpr ((DBG & 4) != 0)
            if (in_loop) return;
#pragma unroll
        for (int i = 0; i < NA; ++i) ga[set][i] = *reinterpret_cast<const f32x4 *>(ap[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i)  // K-step k0 (multiple of 16): block step k0 / 32, k groups (k0 / 8) % 4 + {0, 1}
            gw[set][i] = *reinterpret_cast<const u32x4 *>(wp[i] + (k0 >> 5) * 8192 + ((k0 >> 3) & 3) * 1024);
        if constexpr (LN) {
#pragma unroll
            for (int r = 0; r < NA; ++r) {
                // loaded with the K-step's A data (8 bytes per row, L1-resident): the statistics travel
                // with the staged set across tile switches
                const float2 ms = *reinterpret_cast<const float2 *>(sp[r]);
                st_mean[set][r] = ms.x;
                st_rstd[set][r] = ms.y;
            }
            st_tab[set] = 2 * (k0 + lq * 4);
        }
        st_ln[set] = cur_ln;
        st_sraw[set] = cur_sraw;
    };
    const int st_a = (lq >> 1) * C::SUBA + lrow * 16 + (lq & 1) * 8;
    const int st_b = 2 * C::PLA + (lq & 1) * C::PLB + (lq >> 1) * C::SUBB + lrow * 16;
    auto sstore_a = [&](int set, int i, int stage, bool in_loop = true) {
        if constexpr ((DBG & 2) != 0)
            if (in_loop) return;
        f32x4 v = ga[set][i];
        float s = st_sraw[set];
        if constexpr (LN) {
            // branch-free: tiles that take a itself run the same arithmetic with mean 0, rstd 1, gamma 1,
            // beta 0 ((a - 0) * 1 * 1 + 0 == a)
            f32x4 gb0 = *reinterpret_cast<const f32x4 *>(ln_tab + st_tab[set]);      // g0 b0 g1 b1
            f32x4 gb1 = *reinterpret_cast<const f32x4 *>(ln_tab + st_tab[set] + 4);  // g2 b2 g3 b3
            const bool on = st_ln[set];
            const f32x4 ident = f32x4{1.0f, 0.0f, 1.0f, 0.0f};
            gb0 = on ? gb0 : ident;
            gb1 = on ? gb1 : ident;
            s = on ? s_ln : st_sraw[set];
            const float m = on ? st_mean[set][i] : 0.0f, rs = on ? st_rstd[set][i] : 1.0f;
            v[0] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[0], m), rs), gb0[0]), gb0[1]);
            v[1] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[1], m), rs), gb0[2]), gb0[3]);
            v[2] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[2], m), rs), gb1[0]), gb1[1]);
            v[3] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[3], m), rs), gb1[2]), gb1[3]);
        }
        v *= s;  // exact (power of two)
        const f16x4 h = __builtin_convertvector(v, f16x4);                                // round to nearest
        f16x4 l = h;
        if constexpr ((DBG & 1) == 0)
            l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);  // exact difference
        char *base = smem_c + stage * C::STAGE + st_a + i * RPP * 16;
        *reinterpret_cast<u32x2 *>(base) = __builtin_bit_cast(u32x2, h);
        *reinterpret_cast<u32x2 *>(base + C::PLA) = __builtin_bit_cast(u32x2, l);
    };
    auto sstore_w = [&](int set, int i, int stage, bool in_loop = true) {
        if constexpr ((DBG & 2) != 0)
            if (in_loop) return;
        *reinterpret_cast<u32x4 *>(smem_c + stage * C::STAGE + st_b + i * RPP * 16) = gw[set][i];
    };

    // ---- fragments: A hi (X) / A lo (Y) alternate roles per step; B planes [0] hi, [1] lo --------
    u32x4 fax[MI], fay[MI];
    u32x4 fb[2][NI];
    const int a_frag = hi * C::SUBA + (wm * MI * 32 + r32) * 16;
    const int b_frag = 2 * C::PLA + hi * C::SUBB + (wn * NI * 32 + r32) * 16;
    auto lda = [&](u32x4 (&fa)[MI], int stage, int pl, int mi) {
        fa[mi] = *reinterpret_cast<const u32x4 *>(smem_c + stage * C::STAGE + pl * C::PLA + a_frag + mi * 512);
    };
    auto ldb = [&](int stage, int pl, int ni) {
        fb[pl][ni] = *reinterpret_cast<const u32x4 *>(smem_c + stage * C::STAGE + pl * C::PLB + b_frag + ni * 512);
    };
    f32x16 acc[MI][NI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    zero_acc();
    auto mfma_group = [&](u32x4 (&fa)[MI], int plb) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mi]),
                                                                    __builtin_bit_cast(f16x8, fb[plb][ni]),
                                                                    acc[mi][ni], 0, 0, 0);
    };

    // One K-step (PAR = its parity = its LDS stage; register set PAR^1 holds K-step g+1, set PAR receives
    // the loads of K-step g+2).  On entry X = A hi and fb[1] = W lo of this step (prefetched).
    //   G1  X.Wlo  | loads W hi, Y <- A lo; first half of the split + store of K-step g+1
    //   G2  X.Whi  | second half of the stores                  (every read of stage PAR is issued)
    //   barrier (LDS only)
    //   G3  Y.Whi  | X <- A hi, fb[1] <- W lo of step g+1 (stage PAR^1)
    auto kstep = [&](auto par_c, int k0) {
        constexpr int PAR = decltype(par_c)::value;
        gload(PAR, k0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ldb(PAR, 0, ni);
        mfma_group(fax, 1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) lda(fay, PAR, 1, mi);
#pragma unroll
        for (int i = 0; i < NA; ++i) sstore_a(PAR ^ 1, i, PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(fax, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) sstore_w(PAR ^ 1, i, PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((DBG & 8) == 0) lds_barrier();
        mfma_group(fay, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) lda(fax, PAR ^ 1, 0, mi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ldb(PAR ^ 1, 1, ni);
        __builtin_amdgcn_sched_barrier(0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    set_ptrs(0);
    gload(0, 0, false);
    gload(1, SBK, false);
    if constexpr (LN) __syncthreads();
#pragma unroll
    for (int i = 0; i < NA; ++i) sstore_a(0, i, 0, false);
#pragma unroll
    for (int i = 0; i < NB; ++i) sstore_w(0, i, 0, false);
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) lda(fax, 0, 0, mi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) ldb(0, 1, ni);

    for (int ti = 0; ti < my_tiles; ++ti) {
        const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
        int64_t ebm;
        int ebn;
        tile_coords(lid, nbm, nbn, p.gn, ebm, ebn);
        for (int k = 0; k < nk - 2; k += 2) {
            kstep(P0{}, (k + 2) * SBK);
            kstep(P1{}, (k + 3) * SBK);
        }
        if (ti + 1 < my_tiles) set_ptrs(ti + 1);
        kstep(P0{}, 0);
        kstep(P1{}, SBK);
        // undo the scales: 1 / s_a (LN or raw tile) times the weight row's factor
        const float ainv = 1.0f / ((LN && ebn * TN < hp.ln_cols) ? s_ln : raw_scale(ebm * TM));
        const int col0 = ebn * TN + wn * NI * 32;
        float cs[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = col0 + ni * 32 + r32;
            cs[ni] = ainv * hp.wscale[col < p.N ? col : p.N - 1];
        }
        bool v_done = false;
        if constexpr (FLAGS == 0 && LN && NI % 2 == 0) {
            if (hp.vp && col0 >= hp.v_col0) {  // wave-uniform
                v_done = true;
                float vb = hp.v_bound;
                if (hp.v_bound_dev) vb *= hp.v_bound_dev[(hp.v_rpc > 0 ? (ebm * TM) / hp.v_rpc : 0) * hp.v_bound_stride];
                const float sv = h2_scale_of_bound(vb);
                const int nblk = hp.v_ntok >> 5;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int64_t row = ebm * TM + (wm * MI + mi) * 32;  // first token of this 32-token block
                    if (row < p.M) {
                        const int64_t seq = row / hp.v_ntok;
                        const int blk = (int)(row - seq * hp.v_ntok) >> 5;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const int vcol = col0 - hp.v_col0 + ni * 32;  // column inside V: head vcol / 64, d-half (vcol / 32) & 1
                            unsigned char *dst = hp.vp + ((seq * hp.v_heads + (vcol >> 6)) * nblk + blk) * 8192 +
                                                 ((((vcol >> 5) & 1) * 2 + hi) * 32 + r32) * 16;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                f32x4 pa, pb;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    pa[e] = (acc[mi][ni][8 * j + e] * cs[ni]) * sv;  // the fp32 value the plain epilogue stores,
                                    pb[e] = (acc[mi][ni][8 * j + 4 + e] * cs[ni]) * sv;  // then attn_pack's scaling
                                }
                                const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                                const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                                const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                                *reinterpret_cast<u32x4 *>(dst + j * 2048) =
                                    __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
                                *reinterpret_cast<u32x4 *>(dst + 4096 + j * 2048) =
                                    __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
                            }
                        }
                    }
                }
            }
        }
        if (v_done) {
        } else if (hp.split_col > 0 && col0 >= hp.split_col) {  // wave-uniform: split_col is a multiple of the tile width
            GemmParams q = p;
            q.c = hp.c2;
            q.ldc = hp.ldc2;
            q.N = p.N - hp.split_col;
            if (q.bias) q.bias += hp.split_col;
            x3_epilogue<FLAGS, MI, NI, true>(q, acc, ebm * TM + wm * MI * 32, col0 - hp.split_col, r32, hi, cs);
        } else {
            x3_epilogue<FLAGS, MI, NI, true>(p, acc, ebm * TM + wm * MI * 32, col0, r32, hi, cs);
        }
        zero_acc();
    }
}

// weight row n -> scaled fp16 planes: w' = w * 2^(14 - x) with max|w_n| = m 2^x (m in [0.5, 1)), hi = fp16(w'),
// lo = fp16(w' - hi); scale[n] = 2^(x - 14).  Layout in blocks of 64 rows x 32 k:
// out[n / 64][k / 32][plane][(k / 8) % 4][n % 64][k % 8]  (8 KB per block; the loader's 16 lanes x 16 B of one
// (plane, k-group) are 256 contiguous bytes).  Rows beyond N (up to the next multiple of 64) are zero.
// One wave per row.
__device__ __forceinline__ int64_t h2_w_offset(int n, int k, int plane, int K) {  // in fp16 elements
    return (((int64_t)(n >> 6) * (K >> 5) + (k >> 5)) * 8 + plane * 4 + ((k >> 3) & 3)) * 512 + (n & 63) * 8 + (k & 7);
}

__global__ __launch_bounds__(256) void h2_pack_weight_kernel(const float *__restrict__ w, int64_t ldw, int N, int K,
                                                             _Float16 *__restrict__ out, float *__restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int npad = ((N + 63) / 64) * 64;
    if (n >= npad) return;
    if (n >= N) {
        for (int k = lane; k < K; k += 64) {
            out[h2_w_offset(n, k, 0, K)] = (_Float16)0.0f;
            out[h2_w_offset(n, k, 1, K)] = (_Float16)0.0f;
        }
        return;
    }
    const float *wr = w + (int64_t)n * ldw;
    float mx = 0.0f;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(wr[k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    int x = 14;
    if (mx > 0.0f && mx < 3.0e38f) (void)frexpf(mx, &x);
    const float s = ldexpf(1.0f, 14 - x);
    if (lane == 0) scale[n] = ldexpf(1.0f, x - 14);
    for (int k = lane; k < K; k += 64) {
        const float v = wr[k] * s;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        out[h2_w_offset(n, k, 0, K)] = h;
        out[h2_w_offset(n, k, 1, K)] = l;
    }
}

int g_h2_dbg = 0;  // "h2_dbg": ablation build of the plain 256x256 kernel (measurement only)

template <int FLAGS, typename C, bool LN, int DBG = 0>
static int launch_h2_cfg(H2Params hp, int n_cu, hipStream_t stream) {
    GemmParams &p = hp.g;
    const int lds = 2 * C::STAGE + (LN ? C::LN_TAB : 0);
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_h2_kernel<FLAGS, C, LN, DBG>), lds)) return rc;
    const int64_t nbm = (p.M + C::TM - 1) / C::TM;
    const int nbn = (p.N + C::TN - 1) / C::TN;
    const int64_t nt = nbm * nbn;
    OT_CHECK_ARG(nt < (1ll << 31), "gemm_h2: grid too large");
    p.nbm = (int)nbm;
    p.nbn = nbn;
    p.ntiles = (int)nt;
    int wg_per_cu = (160 * 1024) / lds;
    const int by_waves = 8 / (C::NT / 64);
    if (wg_per_cu > by_waves) wg_per_cu = by_waves;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const int64_t cap = (int64_t)n_cu * wg_per_cu;
    const int grid = (int)(nt < cap ? nt : cap);
    hipLaunchKernelGGL((gemm_h2_kernel<FLAGS, C, LN, DBG>), dim3(grid), dim3(C::NT), lds, stream, hp);
    OT_LAUNCH_CHECK("gemm_h2");
    return OMNITOK_OK;
}

template <int FLAGS, bool LN>
static int launch_h2(H2Params hp, hipStream_t stream) {
    int n_cu = 0;
    if (int rc = current_device_cus(&n_cu)) return rc;
    const GemmParams &p = hp.g;
    auto tiles = [&](int tm, int tn) { return ((p.M + tm - 1) / tm) * (int64_t)((p.N + tn - 1) / tn); };
    int t = g_h2_tile;
    if (t == 0) {
        if (tiles(256, 256) >= 2 * n_cu) t = 1;
        else if (tiles(128, 128) >= n_cu) t = 3;
        else t = 4;
    }
    // a tile must lie inside one clip (one A scale per tile)
    const int64_t rpc = (hp.a_bound_dev && hp.a_rpc > 0) ? hp.a_rpc : 256;
    if ((t == 1 || t == 5) && rpc % 256) t = 3;
    if (t == 6 && rpc % 128) t = 4;
    if (t == 3 && rpc % 128) t = 4;
    if (hp.vp && t == 4) {  // the V-plane epilogue needs whole heads per wave (NI even): 128x128 tiles at least
        if (rpc % 128) {
            set_error("gemm_h2: packed-V output needs rows per clip %% 128 == 0 (got %lld)", (long long)rpc);
            return OMNITOK_ERR_INVALID;
        }
        t = 3;
    }
    // short-K GEMMs with a residual epilogue (to_out: K = 512, 1 GB of traffic for 86 GFLOP): two independent
    // 128x256 workgroups per CU let one tile's residual loads / stores overlap the other's K loop
    // (measured at C3: 0.429 -> 0.380 ms; profiles/r02_h2_tile_experiment.txt)
    if (g_h2_tile == 0 && t == 1 && (FLAGS & OMNITOK_GEMM_RESIDUAL) && !LN && p.K <= 512 && rpc % 128 == 0) t = 6;
    constexpr bool GEGLU = (FLAGS & OMNITOK_GEMM_GEGLU) != 0;
    if constexpr (FLAGS == 0 && !LN) {
        if (t == 1 && g_h2_dbg) {
            using C1 = H2Cfg<2, 4, 4, 2>;
            switch (g_h2_dbg) {
                case 1: return launch_h2_cfg<0, C1, false, 1>(hp, n_cu, stream);
                case 2: return launch_h2_cfg<0, C1, false, 2>(hp, n_cu, stream);
                case 4: return launch_h2_cfg<0, C1, false, 4>(hp, n_cu, stream);
                case 6: return launch_h2_cfg<0, C1, false, 6>(hp, n_cu, stream);
                case 8: return launch_h2_cfg<0, C1, false, 8>(hp, n_cu, stream);
                case 14: return launch_h2_cfg<0, C1, false, 14>(hp, n_cu, stream);
                default: break;
            }
        }
    }
    switch (t) {
        case 1: return launch_h2_cfg<FLAGS, H2Cfg<2, 4, 4, 2>, LN>(hp, n_cu, stream);
        case 3: return launch_h2_cfg<FLAGS, H2Cfg<2, 2, 2, 2>, LN>(hp, n_cu, stream);
        case 5: return launch_h2_cfg<FLAGS, H2Cfg<2, 2, 4, 2>, LN>(hp, n_cu, stream);  // 256x128, 2 WGs per CU
        case 6: return launch_h2_cfg<FLAGS, H2Cfg<2, 2, 2, 4>, LN>(hp, n_cu, stream);  // 128x256, 2 WGs per CU
        default:
            if constexpr (GEGLU) return launch_h2_cfg<FLAGS, H2Cfg<2, 2, 1, 2>, LN>(hp, n_cu, stream);
            else return launch_h2_cfg<FLAGS, H2Cfg<2, 2, 1, 1>, LN>(hp, n_cu, stream);
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_h2_pack_weight(const float *w, int64_t ldw, int N, int K, void *planes, float *scale,
                                      omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(w && planes && scale && N > 0 && K > 0 && K % 32 == 0, "h2_pack_weight: bad arguments (K %% 32)");
    const int npad = ((N + 63) / 64) * 64;
    hipLaunchKernelGGL(h2_pack_weight_kernel, dim3((npad + 3) / 4), dim3(256), 0, stream, w, ldw, N, K,
                       static_cast<_Float16 *>(planes), scale);
    OT_LAUNCH_CHECK("h2_pack_weight");
    return OMNITOK_OK;
}

extern "C" int omnitok_gemm_h2(const float *a, int64_t lda, const void *w_planes, const float *w_scale,
                               const float *bias, const float *residual, int64_t ldr, float *c, int64_t ldc,
                               int64_t M, int N, int K, int flags, int64_t a_rows_per_group, int64_t a_group_stride,
                               int64_t a_group_offset, float a_bound, const float *a_bound_dev, int a_bound_stride,
                               int64_t a_rows_per_clip, const float *ln_stats, const float *ln_gamma, const float *ln_beta, int ln_cols,
                               float ln_bound, float *c2, int64_t ldc2, int split_col, omnitok_stream_t stream_) {
    return omnitok_gemm_h2_vpack(a, lda, w_planes, w_scale, bias, residual, ldr, c, ldc, M, N, K, flags, a_rows_per_group,
                                 a_group_stride, a_group_offset, a_bound, a_bound_dev, a_bound_stride, a_rows_per_clip,
                                 ln_stats, ln_gamma, ln_beta, ln_cols, ln_bound, c2, ldc2, split_col, nullptr, 0, 0, 0, 0.0f,
                                 nullptr, 1, stream_);
}

extern "C" int omnitok_gemm_h2_vpack(const float *a, int64_t lda, const void *w_planes, const float *w_scale,
                                     const float *bias, const float *residual, int64_t ldr, float *c, int64_t ldc,
                                     int64_t M, int N, int K, int flags, int64_t a_rows_per_group, int64_t a_group_stride,
                                     int64_t a_group_offset, float a_bound, const float *a_bound_dev, int a_bound_stride,
                                     int64_t a_rows_per_clip, const float *ln_stats, const float *ln_gamma,
                                     const float *ln_beta, int ln_cols, float ln_bound, float *c2, int64_t ldc2,
                                     int split_col, void *v_planes, int v_col0, int n_tokens, int heads, float v_bound,
                                     const float *v_bound_dev, int v_bound_stride, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(!v_planes || (flags == 0 && ln_stats && v_col0 > 0 && v_col0 % 256 == 0 && v_col0 < N &&
                               (N - v_col0) == heads * 64 && n_tokens > 0 && n_tokens % 32 == 0 && M % n_tokens == 0 &&
                               v_bound > 0.0f && aligned16(v_planes) && a_rows_per_group == 0 &&
                               (!v_bound_dev || a_rows_per_clip > 0)),
                 "gemm_h2: packed-V output needs the fused-LN plain epilogue, v_col0 %% 256 == 0, N - v_col0 == heads * 64, "
                 "whole sequences of n_tokens %% 32 == 0 rows and a positive bound");
    OT_CHECK_ARG(a && w_planes && w_scale && c, "gemm_h2: null pointer");
    OT_CHECK_ARG(split_col == 0 || (c2 && split_col % 256 == 0 && split_col < N && !(flags & (OMNITOK_GEMM_GEGLU | OMNITOK_GEMM_RESIDUAL))),
                 "gemm_h2: split output needs c2, split_col %% 256 == 0 and no GEGLU / residual epilogue");
    OT_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_h2: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return OMNITOK_OK;
    OT_CHECK_ARG(K % 32 == 0 && N % 32 == 0, "gemm_h2: K=%d and N=%d must be multiples of 32", K, N);
    OT_CHECK_ARG(lda % 4 == 0 && aligned16(a) && aligned16(w_planes), "gemm_h2: operands must be 16-byte aligned");
    OT_CHECK_ARG(a_bound > 0.0f, "gemm_h2: an upper bound of |a| is required (fp16 range)");
    OT_CHECK_ARG(!a_bound_dev || a_rows_per_clip <= 0 || (a_rows_per_clip % 64 == 0 && a_rows_per_group == 0),
                 "gemm_h2: rows per clip (%lld) must be a multiple of 64 (tiles may not straddle clips)",
                 (long long)a_rows_per_clip);
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_BIAS) || bias, "gemm_h2: BIAS flag without bias");
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_RESIDUAL) || residual, "gemm_h2: RESIDUAL flag without residual");
    const bool ln = ln_stats != nullptr;
    OT_CHECK_ARG(!ln || (ln_gamma && K <= 512 && ln_cols > 0 && ln_bound > 0.0f && (ln_cols >= N || ln_cols % 256 == 0)),
                 "gemm_h2: fused LayerNorm needs gamma, a bound, K <= 512 and ln_cols a multiple of 256 (or >= N)");
    H2Params hp;
    GemmParams &p = hp.g;
    p.a = a; p.w = nullptr; p.bias = bias; p.residual = residual; p.c = c;
    p.lda = lda; p.ldw = K; p.ldr = ldr; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.a_rpg = a_rows_per_group; p.a_stride = a_group_stride; p.a_off = a_group_offset;
    p.gn = g_gemm_gn > 0 ? g_gemm_gn : 8;
    p.trace = nullptr;
    hp.wpl = w_planes; hp.wscale = w_scale; hp.a_bound = a_bound; hp.a_bound_dev = a_bound_dev;
    hp.a_bound_stride = a_bound_stride > 0 ? a_bound_stride : 1; hp.a_rpc = a_rows_per_clip;
    hp.ln_stats = ln_stats; hp.ln_gamma = ln_gamma; hp.ln_beta = ln_beta; hp.ln_cols = ln ? ln_cols : 0;
    hp.ln_bound = ln_bound;
    hp.c2 = c2; hp.ldc2 = ldc2; hp.split_col = split_col;
    hp.vp = static_cast<unsigned char *>(v_planes); hp.v_col0 = v_col0; hp.v_ntok = n_tokens; hp.v_heads = heads;
    hp.v_bound = v_bound; hp.v_bound_dev = v_bound_dev; hp.v_bound_stride = v_bound_stride > 0 ? v_bound_stride : 1;
    hp.v_rpc = a_rows_per_clip;
#define H2_CASE(F)                                                    \
    case F:                                                           \
        return ln ? launch_h2<F, true>(hp, stream) : launch_h2<F, false>(hp, stream);
    switch (flags) {
        H2_CASE(0)
        H2_CASE(OMNITOK_GEMM_BIAS)
        H2_CASE(OMNITOK_GEMM_RESIDUAL)
        H2_CASE(OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL)
        case OMNITOK_GEMM_GEGLU:
            OT_CHECK_ARG(N % 64 == 0, "gemm_h2: GEGLU packed width %d must be a multiple of 64", N);
            return ln ? launch_h2<OMNITOK_GEMM_GEGLU, true>(hp, stream)
                      : launch_h2<OMNITOK_GEMM_GEGLU, false>(hp, stream);
        default:
            set_error("gemm_h2: unsupported epilogue flags %d", flags);
            return OMNITOK_ERR_INVALID;
    }
#undef H2_CASE
}
