// fp32 GEMM  c = a . w^T (+ epilogue) on the bf16 matrix cores by exact 3-way operand splitting,
// with the split done INSIDE the kernel while the operands are staged to LDS ("x3" GEMM).
//
// Why: the fp32-input MFMA runs at 1/16 of the bf16 MFMA rate (157 vs 2500 TF) and the path's
// GEMMs (reference attention.py:164,167,271,287,386-393; omnitokenizer.py:810,819,1007,1013) were
// at 0.83-0.88 of that roof.  Every fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8
// significand bits, truncation split: a0 = top 16 bits of a, a1 = top 16 bits of a - a0,
// a2 = a - a0 - a1), so a*b = sum_ij a_i*b_j with every bf16 x bf16 product exact in fp32.  The six
// products with weight >= 2^-16 are evaluated on v_mfma_f32_32x32x16_bf16 (fp32 accumulate); the three
// dropped ones are below 2^-24 |a b|, i.e. below half an fp32 ulp of the product.  6 x 32 = 192
// matrix-pipe cycles per 32x32x16 block instead of 8 x 64 = 512: a 417 TF "fp32-equivalent" roof.
//
// What round 1's experiment (planes split by a separate pass, 128x128 tiles) taught: six bf16 planes
// are 6 B per element of global traffic and a 128x128 tile then needs 32 B/clk/CU of operand
// delivery; it saturated at ~12.  Here the operands stay fp32 in global memory (4 B per element, no
// split pass, no change to any producer), each thread splits the float4 it staged with ~5.5 VALU
// per element between the MFMAs, and the tile is up to 256x256 (10.7 B/clk/CU at full rate).
//
// Structure (one persistent workgroup per CU walks the XCD-aware tile list of gemm_common.h):
//  * WGM x WGN waves, each an (MI*32) x (NI*32) sub-tile; BK = 16 per K-step; two LDS stages;
//  * LDS image of a stage: per operand and plane two sub-planes [k half][row][8 bf16] with 16 B per
//    row, so a fragment (MFMA operand: lane = row + 32 * k half, 8 bf16) is ONE ds_read_b128 and the
//    16-lane groups of ds_read_b128 hit 16 distinct 16-B slots; sub-planes are 64 B apart mod 128 so
//    the ds_write_b64 of the loader (4 lanes = one 64-B row segment) are conflict-free too;
//  * K-step g: six MFMA groups (a0b2 a0b1 a0b0 a1b0 a1b1 a2b0; every element sees the same sequence
//    whatever the tile shape -> results do not depend on the tiling / batch size); the A fragments
//    live in two alternating register sets and are fetched one group ahead; between the MFMAs the
//    thread splits + stores K-step g+1 (loaded one step ago) into the other stage and issues the
//    global loads of K-step g+2; ONE LDS-only barrier per K-step (before the last group, whose cover
//    prefetches the first fragments of step g+1);
//  * optional LayerNorm fused into the A staging (mean / rstd per row from a stats pass, gamma/beta
//    from an LDS table): the LN output never exists in HBM, and a column range [0, ln_cols) of one
//    launch can take LN(x) while the rest takes x (Q from LN(x), K/V from x: attention.py:404-412).
#include "gemm_x_common.h"

#include <type_traits>

namespace omnitok {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

int g_x3_tile = 0;  // "x3_tile": 0 auto, 1 256x256, 2 256x128, 3 128x128, 4 64x128, 5 256x128 x 4 waves (2 WG/CU),
                    // 6 128x256 x 4 waves, 7 256x256 one staged set (A/B measurements)

struct X3Params {
    GemmParams g;
    const float *ln_stats;   // [rows][2] (mean, rstd) indexed by the physical A row, or null
    const float *ln_gamma;   // [K]
    const float *ln_beta;    // [K] or null (= 0)
    int ln_cols;             // output columns [0, ln_cols) take LN(a), the rest takes a
    float *c2;               // output columns [split_col, N) go to c2 (row stride ldc2) instead of c
    int64_t ldc2;            //   (Q | K,V of the merged to_q / to_kv launch as two dense tensors)
    int split_col;           // 0: everything to c
};

// ONE: a single staged register set (the global loads of K-step g+1 are issued at the start of step g
// and split/stored in its second half) instead of two sets with a two-step prefetch distance
template <int WGM_, int WGN_, int MI_, int NI_, bool ONE_ = false>
struct X3Cfg {
    static constexpr int WGM = WGM_, WGN = WGN_, MI = MI_, NI = NI_;
    static constexpr bool ONE = ONE_;
    static constexpr int NT = 64 * WGM * WGN;
    static constexpr int TM = WGM * MI * 32, TN = WGN * NI * 32;
    static constexpr int RPP = NT / 4;  // tile rows per loader pass: 4 lanes x 16 B = one 64-B row segment
    static constexpr int NA = TM / RPP, NB = TN / RPP, NP = NA + NB;
    static constexpr int SUBA = TM * 16 + 64, SUBB = TN * 16 + 64;  // one [k half] sub-plane
    static constexpr int PLA = 2 * SUBA, PLB = 2 * SUBB;
    static constexpr int STAGE = 3 * PLA + 3 * PLB;
    static constexpr int LN_TAB = 2 * 512 * 4;  // gamma|beta pairs, K <= 512
    static_assert(TM % RPP == 0 && TN % RPP == 0, "loader passes");
};

// v (4 consecutive k) -> three planes of 4 bf16; v == p0 + p1 + p2 exactly
__device__ __forceinline__ void split3x4(const f32x4 v, u32x2 &p0, u32x2 &p1, u32x2 &p2) {
    unsigned b0[4], b1[4], b2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];
        b0[e] = __float_as_uint(x);
        const float r1 = __fsub_rn(x, __uint_as_float(b0[e] & 0xFFFF0000u));   // exact
        b1[e] = __float_as_uint(r1);
        const float r2 = __fsub_rn(r1, __uint_as_float(b1[e] & 0xFFFF0000u));  // exact, <= 8 significant bits
        b2[e] = __float_as_uint(r2);
    }
    // v_perm_b32: [hi16(odd element) : hi16(even element)]
    p0[0] = __builtin_amdgcn_perm(b0[1], b0[0], 0x07060302u);
    p0[1] = __builtin_amdgcn_perm(b0[3], b0[2], 0x07060302u);
    p1[0] = __builtin_amdgcn_perm(b1[1], b1[0], 0x07060302u);
    p1[1] = __builtin_amdgcn_perm(b1[3], b1[2], 0x07060302u);
    p2[0] = __builtin_amdgcn_perm(b2[1], b2[0], 0x07060302u);
    p2[1] = __builtin_amdgcn_perm(b2[3], b2[2], 0x07060302u);
}

// DBG (measurement builds only, wrong results): 1 skip the split arithmetic, 2 skip the LDS stores of the
// K loop, 4 skip the global loads of the K loop, 8 skip the per-step barrier
template <int FLAGS, typename C, bool LN, int DBG = 0>
__global__ __launch_bounds__(C::NT, 2) void gemm_x3_kernel(X3Params xp) {
    constexpr int MI = C::MI, NI = C::NI, NA = C::NA, NB = C::NB, NP = C::NP;
    constexpr int TM = C::TM, TN = C::TN, RPP = C::RPP, SBK = 16;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const GemmParams &p = xp.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nk = p.K / SBK;
    if ((int)blockIdx.x >= p.ntiles) return;
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nbm = (int)((p.M + TM - 1) / TM), nbn = (p.N + TN - 1) / TN;
    long long dbg_c0 = 0, dbg_w0 = 0;
    if constexpr ((DBG & 16) != 0) {  // shader-clock / 100 MHz wall-clock stamps -> effective clock
        dbg_c0 = clock64();
        dbg_w0 = wall_clock64();
    }

    // LayerNorm table in LDS (after the two stages): gamma|beta pairs
    float *ln_tab = reinterpret_cast<float *>(smem_c + 2 * C::STAGE);
    if constexpr (LN) {
        for (int k = tid; k < p.K; k += C::NT) {
            ln_tab[2 * k] = xp.ln_gamma[k];
            ln_tab[2 * k + 1] = xp.ln_beta ? xp.ln_beta[k] : 0.0f;
        }
    }

    // ---- loader: thread -> k quad (tid & 3) of rows (tid >> 2) + RPP * i --------------------
    const int lrow = tid >> 2, lq = tid & 3;
    const float *ap[NA];
    const float *wp[NB];
    const float *sp[NA];               // (mean, rstd) of the rows the load pointers address
    bool cur_ln = false;               // does the tile the load pointers address take LN(a)?
    auto set_ptrs = [&](int i) {
        const int lid = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, p.ntiles);
        int64_t tbm;
        int tbn;
        tile_coords(lid, nbm, nbn, p.gn, tbm, tbn);
        const bool ln_on = LN && tbn * TN < xp.ln_cols;
        cur_ln = ln_on;
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            int64_t gr = tbm * TM + lrow + RPP * r;
            if (gr > p.M - 1) gr = p.M - 1;
            int64_t ar = gr;
            if (p.a_rpg > 0) ar = (gr / p.a_rpg) * p.a_stride + p.a_off + (gr % p.a_rpg);
            ap[r] = p.a + ar * p.lda + lq * 4;
            if constexpr (LN) sp[r] = xp.ln_stats + 2 * ar;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            int wr = tbn * TN + lrow + RPP * r;
            if (wr > p.N - 1) wr = p.N - 1;
            wp[r] = p.w + (int64_t)wr * p.ldw + lq * 4;
        }
    };
    // two register sets of staged K-steps (set k&1 holds K-step k), plus what the LN transform of
    // that K-step needs (the rows' statistics and the k offset into the table)
    f32x4 gs[2][NP];
    float st_mean[2][NA], st_rstd[2][NA];
    int st_tab[2];
    bool st_ln[2];
    auto gload = [&](int set, int piece, int k0) {
        gs[set][piece] = *reinterpret_cast<const f32x4 *>((piece < NA ? ap[piece] : wp[piece - NA]) + k0);
    };
    auto gload_meta = [&](int set, int k0) {
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
            st_ln[set] = cur_ln;
        }
    };
    const int st_a = (lq >> 1) * C::SUBA + lrow * 16 + (lq & 1) * 8;
    const int st_b = 3 * C::PLA + (lq >> 1) * C::SUBB + lrow * 16 + (lq & 1) * 8;
    auto sstore = [&](int set, int piece, int stage, bool in_loop = true) {
        if constexpr ((DBG & 2) != 0)
            if (in_loop) return;
        f32x4 v = gs[set][piece];
        if constexpr (LN) {
            if (piece < NA) {
                f32x4 gb0 = *reinterpret_cast<const f32x4 *>(ln_tab + st_tab[set]);      // g0 b0 g1 b1
                f32x4 gb1 = *reinterpret_cast<const f32x4 *>(ln_tab + st_tab[set] + 4);  // g2 b2 g3 b3
                if (!st_ln[set]) {  // tile-uniform: this column range takes a itself ((a - 0) * 1 * 1 + 0)
                    gb0 = f32x4{1.0f, 0.0f, 1.0f, 0.0f};
                    gb1 = gb0;
                }
                const float m = st_ln[set] ? st_mean[set][piece] : 0.0f, rs = st_ln[set] ? st_rstd[set][piece] : 1.0f;
                v[0] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[0], m), rs), gb0[0]), gb0[1]);
                v[1] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[1], m), rs), gb0[2]), gb0[3]);
                v[2] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[2], m), rs), gb1[0]), gb1[1]);
                v[3] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[3], m), rs), gb1[2]), gb1[3]);
            }
        }
        u32x2 p0, p1, p2;
        if constexpr ((DBG & 1) != 0) {
            p0[0] = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);
            p0[1] = __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07060302u);
            p1 = p0;
            p2 = p0;
        } else {
            split3x4(v, p0, p1, p2);
        }
        char *base = smem_c + stage * C::STAGE +
                     (piece < NA ? st_a + piece * RPP * 16 : st_b + (piece - NA) * RPP * 16);
        const int pl = piece < NA ? C::PLA : C::PLB;
        *reinterpret_cast<u32x2 *>(base) = p0;
        *reinterpret_cast<u32x2 *>(base + pl) = p1;
        *reinterpret_cast<u32x2 *>(base + 2 * pl) = p2;
    };

    // ---- fragments --------------------------------------------------------------------------
    u32x4 fax[MI], fay[MI];      // two alternating A fragment sets
    u32x4 fb[3][NI];             // the three B planes of the current K-step
    const int a_frag = hi * C::SUBA + (wm * MI * 32 + r32) * 16;
    const int b_frag = 3 * C::PLA + hi * C::SUBB + (wn * NI * 32 + r32) * 16;
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
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[mi]),
                                                                     __builtin_bit_cast(bf16x8, fb[plb][ni]),
                                                                     acc[mi][ni], 0, 0, 0);
    };

    // One K-step.  PAR = parity of the step: LDS stage PAR holds its operands, register set PAR^1
    // holds K-step g+1 (stored into stage PAR^1 here), register set PAR is free and receives the
    // global loads of K-step g+2 (at k offset k0 of the current load pointers).
    // On entry: X = a0 of this step, fb[2] = b2 of this step (prefetched).  Groups:
    //   G1 X.b2   | loads b1, b0; Y <- a1
    //   G2 X.b1   |
    //   G3 X.b0   | (X dead afterwards)
    //   G4 Y.b0   | X <- a2
    //   G5 Y.b1   | (Y dead afterwards)          -- every read of stage PAR is issued by now
    //   barrier (LDS only)
    //   G6 X.b0   | Y <- a0 of step g+1, fb[2] <- b2 of step g+1 (stage PAR^1)
    // so the roles of X and Y swap every step (the caller alternates them).
    auto kstep = [&](auto par_c, u32x4 (&X)[MI], u32x4 (&Y)[MI], int k0) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int LS = C::ONE ? 0 : PAR;        // register set that receives this step's global loads
        constexpr int SS = C::ONE ? 0 : (PAR ^ 1);  // register set that is split + stored in this step
        // global loads first (ONE: K-step g+1, consumed in the second half of this step; else K-step
        // g+2, a whole K-step of latency cover)
        if constexpr ((DBG & 4) == 0) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) gload(LS, pc, k0);
        }
        gload_meta(LS, k0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ldb(PAR, 1, ni);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ldb(PAR, 0, ni);
        mfma_group(X, 2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) lda(Y, PAR, 1, mi);
        __builtin_amdgcn_sched_barrier(0);
        // split + store of K-step g+1: spread over groups 2..5, or (ONE) over groups 4..5
        constexpr int S0 = 0, S1 = C::ONE ? 0 : (NP + 3) / 4, S2 = C::ONE ? 0 : (2 * NP + 3) / 4,
                      S3 = C::ONE ? (NP + 1) / 2 : (3 * NP + 3) / 4;
        mfma_group(X, 1);
#pragma unroll
        for (int pc = S0; pc < S1; ++pc) sstore(SS, pc, PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(X, 0);
#pragma unroll
        for (int pc = S1; pc < S2; ++pc) sstore(SS, pc, PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Y, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) lda(X, PAR, 2, mi);
#pragma unroll
        for (int pc = S2; pc < S3; ++pc) sstore(SS, pc, PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Y, 1);
#pragma unroll
        for (int pc = S3; pc < NP; ++pc) sstore(SS, pc, PAR ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((DBG & 8) == 0) lds_barrier();
        mfma_group(X, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) lda(Y, PAR ^ 1, 0, mi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ldb(PAR ^ 1, 2, ni);
        __builtin_amdgcn_sched_barrier(0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // prologue: K-step 0 -> register set 0 -> LDS stage 0 (two sets: K-step 1 -> register set 1)
    set_ptrs(0);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) gload(0, pc, 0);
    gload_meta(0, 0);
    if constexpr (!C::ONE) {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) gload(1, pc, SBK);
        gload_meta(1, SBK);
    }
    if constexpr (LN) __syncthreads();  // the table
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) sstore(0, pc, 0, false);
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) lda(fax, 0, 0, mi);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) ldb(0, 2, ni);

    for (int ti = 0; ti < my_tiles; ++ti) {
        if constexpr (C::ONE) {
            for (int k = 0; k < nk - 2; k += 2) {
                kstep(P0{}, fax, fay, (k + 1) * SBK);
                kstep(P1{}, fay, fax, (k + 2) * SBK);
            }
            kstep(P0{}, fax, fay, (nk - 1) * SBK);
            // the last K-step fetches K-step 0 of the next tile (of this tile again when it is the last)
            if (ti + 1 < my_tiles) set_ptrs(ti + 1);
            kstep(P1{}, fay, fax, 0);
        } else {
            for (int k = 0; k < nk - 2; k += 2) {
                kstep(P0{}, fax, fay, (k + 2) * SBK);
                kstep(P1{}, fay, fax, (k + 3) * SBK);
            }
            // the last two K-steps fetch K-steps 0 and 1 of the next tile (of this tile again when it is
            // the last: the stream re-fetches valid data into idle buffers)
            if (ti + 1 < my_tiles) set_ptrs(ti + 1);
            kstep(P0{}, fax, fay, 0);
            kstep(P1{}, fay, fax, SBK);
        }
        const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
        int64_t ebm;
        int ebn;
        tile_coords(lid, nbm, nbn, p.gn, ebm, ebn);
        {
            const int col0 = ebn * TN + wn * NI * 32;
            if (xp.split_col > 0 && col0 >= xp.split_col) {  // wave-uniform: split_col is a multiple of the tile width
                GemmParams q = p;
                q.c = xp.c2;
                q.ldc = xp.ldc2;
                q.N = p.N - xp.split_col;
                if (q.bias) q.bias += xp.split_col;
                x3_epilogue<FLAGS, MI, NI>(q, acc, ebm * TM + wm * MI * 32, col0 - xp.split_col, r32, hi);
            } else {
                x3_epilogue<FLAGS, MI, NI>(p, acc, ebm * TM + wm * MI * 32, col0, r32, hi);
            }
        }
        zero_acc();
    }
    if constexpr ((DBG & 16) != 0) {
        if (p.trace && tid == 0 && blockIdx.x < 256) {
            p.trace[2 * blockIdx.x] = clock64() - dbg_c0;
            p.trace[2 * blockIdx.x + 1] = wall_clock64() - dbg_w0;
        }
    }
}

extern long long *g_gemm_trace;
int g_x3_dbg = 0;  // "x3_dbg": ablation build of the plain 256x256 kernel (measurement only)

template <int FLAGS, typename C, bool LN, int DBG = 0>
static int launch_x3_cfg(X3Params xp, int n_cu, hipStream_t stream) {
    GemmParams &p = xp.g;
    const int lds = 2 * C::STAGE + (LN ? C::LN_TAB : 0);
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_x3_kernel<FLAGS, C, LN, DBG>), lds)) return rc;
    const int64_t nbm = (p.M + C::TM - 1) / C::TM;
    const int nbn = (p.N + C::TN - 1) / C::TN;
    const int64_t nt = nbm * nbn;
    OT_CHECK_ARG(nt < (1ll << 31), "gemm_x3: grid too large");
    p.nbm = (int)nbm;
    p.nbn = nbn;
    p.ntiles = (int)nt;
    int wg_per_cu = (160 * 1024) / lds;
    const int by_waves = 8 / (C::NT / 64);  // at most two waves per SIMD (register budget of the kernel)
    if (wg_per_cu > by_waves) wg_per_cu = by_waves;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const int64_t cap = (int64_t)n_cu * wg_per_cu;
    const int grid = (int)(nt < cap ? nt : cap);
    hipLaunchKernelGGL((gemm_x3_kernel<FLAGS, C, LN, DBG>), dim3(grid), dim3(C::NT), lds, stream, xp);
    OT_LAUNCH_CHECK("gemm_x3");
    return OMNITOK_OK;
}

template <int FLAGS, bool LN>
static int launch_x3(X3Params xp, hipStream_t stream) {
    int n_cu = 0;
    if (int rc = current_device_cus(&n_cu)) return rc;
    const GemmParams &p = xp.g;
    // tile choice by how many tiles the problem offers per CU (every tile shape performs the same
    // per-element arithmetic, so the choice never changes a result)
    auto tiles = [&](int tm, int tn) { return ((p.M + tm - 1) / tm) * (int64_t)((p.N + tn - 1) / tn); };
    int t = g_x3_tile;
    if (t == 0) {
        if (tiles(256, 256) >= 2 * n_cu) t = 1;
        else if (tiles(128, 128) >= n_cu) t = 3;
        else t = 4;
    }
    constexpr bool GEGLU = (FLAGS & OMNITOK_GEMM_GEGLU) != 0;
    if constexpr (FLAGS == 0 && !LN) {
        if (t == 1 && g_x3_dbg) {
            using C1 = X3Cfg<2, 4, 4, 2>;
            switch (g_x3_dbg) {
                case 1: return launch_x3_cfg<0, C1, false, 1>(xp, n_cu, stream);
                case 2: return launch_x3_cfg<0, C1, false, 2>(xp, n_cu, stream);
                case 4: return launch_x3_cfg<0, C1, false, 4>(xp, n_cu, stream);
                case 6: return launch_x3_cfg<0, C1, false, 6>(xp, n_cu, stream);
                case 8: return launch_x3_cfg<0, C1, false, 8>(xp, n_cu, stream);
                case 14: return launch_x3_cfg<0, C1, false, 14>(xp, n_cu, stream);
                case 16: return launch_x3_cfg<0, C1, false, 16>(xp, n_cu, stream);
                case 30: return launch_x3_cfg<0, C1, false, 30>(xp, n_cu, stream);
                default: break;
            }
        }
    }
    switch (t) {
        case 1: return launch_x3_cfg<FLAGS, X3Cfg<2, 4, 4, 2>, LN>(xp, n_cu, stream);
        case 2: return launch_x3_cfg<FLAGS, X3Cfg<4, 2, 2, 2>, LN>(xp, n_cu, stream);
        case 3: return launch_x3_cfg<FLAGS, X3Cfg<2, 2, 2, 2>, LN>(xp, n_cu, stream);
        case 5: return launch_x3_cfg<FLAGS, X3Cfg<2, 2, 4, 2, true>, LN>(xp, n_cu, stream);
        case 6: return launch_x3_cfg<FLAGS, X3Cfg<2, 2, 2, 4, true>, LN>(xp, n_cu, stream);
        case 7: return launch_x3_cfg<FLAGS, X3Cfg<2, 4, 4, 2, true>, LN>(xp, n_cu, stream);
        default:
            if constexpr (GEGLU) return launch_x3_cfg<FLAGS, X3Cfg<2, 2, 1, 2>, LN>(xp, n_cu, stream);
            else return launch_x3_cfg<FLAGS, X3Cfg<2, 2, 1, 1>, LN>(xp, n_cu, stream);
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_gemm_x3(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                               const float *residual, int64_t ldr, float *c, int64_t ldc, int64_t M, int N, int K,
                               int flags, int64_t a_rows_per_group, int64_t a_group_stride, int64_t a_group_offset,
                               const float *ln_stats, const float *ln_gamma, const float *ln_beta, int ln_cols,
                               float *c2, int64_t ldc2, int split_col, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(a && w && c, "gemm_x3: null pointer");
    OT_CHECK_ARG(split_col == 0 || (c2 && split_col % 256 == 0 && split_col < N && !(flags & (OMNITOK_GEMM_GEGLU | OMNITOK_GEMM_RESIDUAL))),
                 "gemm_x3: split output needs c2, split_col %% 256 == 0 and no GEGLU / residual epilogue");
    OT_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_x3: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return OMNITOK_OK;
    OT_CHECK_ARG(K % 32 == 0, "gemm_x3: K=%d must be a multiple of 32 (pad the weight)", K);
    OT_CHECK_ARG(N % 32 == 0, "gemm_x3: N=%d must be a multiple of 32", N);
    OT_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && aligned16(a) && aligned16(w),
                 "gemm_x3: operands must be 16-byte aligned with ld %% 4 == 0");
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_BIAS) || bias, "gemm_x3: BIAS flag without bias");
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_RESIDUAL) || residual, "gemm_x3: RESIDUAL flag without residual");
    const bool ln = ln_stats != nullptr;
    OT_CHECK_ARG(!ln || (ln_gamma && K <= 512 && ln_cols > 0 && (ln_cols >= N || ln_cols % 256 == 0)),
                 "gemm_x3: fused LayerNorm needs gamma, K <= 512 and ln_cols a multiple of 256 (or >= N)");
    X3Params xp;
    GemmParams &p = xp.g;
    p.a = a; p.w = w; p.bias = bias; p.residual = residual; p.c = c;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.a_rpg = a_rows_per_group; p.a_stride = a_group_stride; p.a_off = a_group_offset;
    p.gn = 8;
    p.trace = g_gemm_trace;
    xp.ln_stats = ln_stats; xp.ln_gamma = ln_gamma; xp.ln_beta = ln_beta; xp.ln_cols = ln ? ln_cols : 0;
    xp.c2 = c2; xp.ldc2 = ldc2; xp.split_col = split_col;
#define X3_CASE(F)                                                    \
    case F:                                                           \
        return ln ? launch_x3<F, true>(xp, stream) : launch_x3<F, false>(xp, stream);
    switch (flags) {
        X3_CASE(0)
        X3_CASE(OMNITOK_GEMM_BIAS)
        X3_CASE(OMNITOK_GEMM_RESIDUAL)
        X3_CASE(OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL)
        case OMNITOK_GEMM_GEGLU:
            OT_CHECK_ARG(N % 64 == 0, "gemm_x3: GEGLU packed width %d must be a multiple of 64", N);
            return ln ? launch_x3<OMNITOK_GEMM_GEGLU, true>(xp, stream)
                      : launch_x3<OMNITOK_GEMM_GEGLU, false>(xp, stream);
        default:
            set_error("gemm_x3: unsupported epilogue flags %d", flags);
            return OMNITOK_ERR_INVALID;
    }
#undef X3_CASE
}
