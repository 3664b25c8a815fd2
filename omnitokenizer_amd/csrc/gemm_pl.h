// "pl" GEMM: c = a . w^T on the fp16 matrix cores with BOTH operands already stored as fp16 hi|lo planes
// (2-way exact-scale split of fp32, see gemm_h2.hip for the arithmetic) in 8 KiB blocks of 64 rows x 32 k:
//     block[row / 64][k / 32] = [plane (hi, lo)][k group (k / 8) % 4][row % 64][k % 8]   fp16
// so that one (plane, k group) of a block is 1 KiB of contiguous memory = one global_load_lds_dwordx4 of a wave,
// and the same bytes in LDS are MFMA fragments (lane (r32, half) reads row r32, k group `half`: 16 B).
//
// The K loop contains nothing but LDS-DMA issues, ds_read_b128 and MFMAs: no VALU work on the operands, no ds_write.
// Producers write the activation planes (LayerNorm / GEGLU / attention epilogues, gemm_pl.hip: omnitok_pl_pack_rows).
//
// Orientation.  The WEIGHT fragments go to the MFMA's a slot and the ACTIVATION fragments to the b slot, i.e. a
// 32x32 accumulator block holds out^T: lane (r32, half) owns ONE token row m = r32 and 16 output columns.  The weight
// rows are stored permuted inside every group of 32 (pl_perm) so that those 16 columns are two runs of 8 consecutive n:
//     registers 0..7  -> n0 + half * 8 + j,     registers 8..15 -> n0 + 16 + half * 8 + j
// A lane therefore stores 32 contiguous bytes per run (fp32 output), or exactly one 16-byte fp16 chunk per plane of
// the NEXT GEMM's activation operand (k group = run * 2 + half): epilogues write plane blocks with 512-byte
// contiguous runs per half wave and no cross-lane traffic.  SWAP = true exchanges the slots (lane = column n,
// registers = 16 token rows: the layout the packed-V epilogue of the attention kernel wants).
//
// Structure: wave tile 64 n x 128 m (2 x 4 accumulator blocks = 128 registers), WN x WM waves per workgroup, K steps
// of 16, a ring of R LDS stages filled two steps ahead, one barrier per step, three MFMA groups per step
// (W lo . A hi | W hi . A hi | W hi . A lo) with the fragment reads of a group issued one group earlier.
#pragma once
#include "gemm_x_common.h"
#include "h2_common.h"

namespace omnitok {

typedef __attribute__((address_space(3))) void pl_lds_t;
typedef __attribute__((address_space(1))) const void pl_glob_t;

enum PlEpi {
    PL_F32 = 0,         // fp32 row-major output (+ bias, + residual)
    PL_GEGLU = 1,       // GEGLU of (value | gate) 32-column pairs -> fp16 planes of the hidden (next GEMM's operand)
    PL_ROWLN = 2,       // full-row tiles (TN == N): fp32 output (+ bias, + residual) AND LayerNorm(out) as planes
    PL_VPACK = 3,       // SWAP only: V columns into the attention kernel's packed fp16 planes (attn_h2.hip)
    PL_QKPACK = 4,      // Q | K columns (one head per wave): RoPE + l2norm + scales -> the attention kernel's packed planes
    PL_UNPATCH = 5,     // fp32 (+ bias) scattered as pixels: the un-patchify Rearrange of to_pixels fused into the store
    // The temporal stage without its q|k|v round trip (T' == 5; reference attention.py:402-486, is_spatial = False): rows ordered
    // [tile of 64 sequences][32-sequence half][time step][sequence], so that the five time steps of a sequence are the five row
    // blocks of ONE lane pair of a wave (PlCfg<2, 2, R, D, 0, 2, 5>: 320 x 128 tiles, wave tile 2 x 5 accumulator blocks = 160
    // registers, one wave per SIMD, 7 DMA pieces per wave and K step)
    PL_TSCORE = 6,      // tile = one head [q_h | k_h]: LayerNorm fold, l2norm, scales; k to the q wave through LDS; causal scores, softmax -> P
    PL_TPV = 7,         // tile = V of two heads: o_t = sum_s P[t][s] v_s in-lane -> the out-projection's operand planes (token order)
};

struct PlParams {
    // activation planes; tiles whose first column is >= a_split_n read a2 / a2_scale instead (merged q | k launch:
    // Q from LayerNorm(x), K from the raw x)
    const unsigned char *a, *a2;
    const float *a_scale, *a2_scale;  // [M] per-row factors that undo the activation scaling, or null
    float a_scale_const, a2_scale_const;  // multiplied in (1 when the per-row array carries everything)
    int a_split_n;                    // 0: no second operand
    const unsigned char *w;           // weight planes (rows permuted, pl_perm)
    const float *w_scale;             // [N]
    const float *bias;                // [N] or null
    const float *residual;
    int64_t ldr;
    float *c;
    int64_t ldc;
    float *c2;                        // columns >= c_split_n go to c2 (row stride ldc2), column index minus c_split_n
    int64_t ldc2;
    int c_split_n;
    // plane outputs (PL_GEGLU: the hidden; PL_ROWLN: LayerNorm(out))
    unsigned char *cp;
    int cp_kblocks;                   // K / 32 of the consumer
    float cp_scale;                   // power of two applied before the split (from a static bound)
    const float *ln_gamma, *ln_beta;  // PL_ROWLN
    float ln_eps;
    // LayerNorm folded into the weight (PL_F32, PL_QKPACK, PL_VPACK): the operand holds the CENTRED rows x - mean(x)
    // (omnitok_stats_pack), the weight rows of the columns n < fold_cols were multiplied by gamma, and the epilogue finishes
    //   n <  fold_cols:  LayerNorm(x) . W^T = rstd_m ((x - mean_m) . (W o gamma)^T) + b_n,      b = W beta
    //   n >= fold_cols:  x . W^T           = (x - mean_m) . W^T + mean_m u_n,                   u_n = sum_k W[n][k]
    // -- the reference's own order for Q (attention.py:73-80: centre first), and no cancellation for K / V (reference
    // attention.py:404-412: Q from LN(x), K / V from the raw x)
    const float *fold_stats;          // [M][2] (mean, rstd), or null
    const float *fold_b;              // [fold_cols] or null
    const float *fold_u;              // [N] row sums of the weight (used for n >= fold_cols), or null when fold_cols >= N
    int fold_cols;
    // PL_QKPACK (reference attention.py:417-437): columns [0, qk_k0) are Q, [qk_k0, N) are K; 64 columns = one head
    unsigned char *qp, *kp;
    int qk_k0, qk_ntok, qk_heads;
    const float *cosT, *sinT;         // [n_tokens][32] or null
    const float *q_scale, *k_scale;   // [64]
    float q_mul, sq, sk;              // SDPA scale folded into q; power-of-two operand scales of q and k
    // PL_VPACK
    unsigned char *vp;
    int v_ntok, v_heads;
    float v_bound;
    const float *v_bound_dev;
    int v_bound_stride;
    int64_t v_rpc;
    int64_t M;
    int N, K;
    int nk;                           // K steps of 16 actually run: ceil(k_valid / 16) <= K / 16 (the rest of K is zero padding)
    int nbm, nbn, ntiles, gn;
    // A launch may cover only the row tiles [bm0, bm0 + nbm) of the problem (r06 tail schedule: full rounds of tall tiles, then the
    // remaining rows on thin ones); every address is formed from the ABSOLUTE row tile, so the pointers stay those of the whole problem.
    // row_begin / row_end are host-side (launch_pl_cfg turns them into bm0 / nbm in units of the configuration's tile height).
    int bm0;
    int64_t row_begin, row_end;
    // operand row map (a_rpg == 0: identity): the frame groups of the token tensor
    int64_t a_rpg, a_gstride, a_goff;
    int up_C, up_F, up_H, up_W, up_f0, up_t, up_pt, up_p;  // PL_UNPATCH
    // PL_TSCORE / PL_TPV: P buffer [sequence][head][40 floats]: per query step t 8 floats = e[t][0..4], 1 / sum_s e[t][s], 0, 0
    float *tp;
    int t_nseq, t_heads, t_seqs_per_clip;
    const float *t_alibi;             // [heads] ALiBi slopes (legacy attention) or null
    float t_scale;                    // SDPA scale folded into q
    float *t_out_scale;               // PL_TPV: [token row] factor that undoes the plane scale (a_scale of the out-projection)
    long long *cycles;                // measurement: s_memtime span of workgroup 0 (null = off)
    int stagger;                      // start delay step in ~1 us units ("pl_stagger"); workgroup phase = (id / 8) % 8
};

// logical row (inside a group of 32) stored at physical position i: the MFMA a-slot row i lands in lane half
// (i >> 2) & 1, register (i >> 3) * 4 + (i & 3)
__host__ __device__ __forceinline__ int pl_perm(int i) {
    const int half = (i >> 2) & 1, r = (i >> 3) * 4 + (i & 3);
    return (r >> 3) * 16 + half * 8 + (r & 7);
}

__host__ __device__ __forceinline__ int64_t pl_offset(int64_t row, int k, int plane, int K) {  // in fp16 elements
    return (((row >> 6) * (K >> 5) + (k >> 5)) * 8 + plane * 4 + ((k >> 3) & 3)) * 512 + (row & 63) * 8 + (k & 7);
}

// s_waitcnt vmcnt(n) only: gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define PL_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
// wait until at most min(ahead, N) K steps of PPWv pieces each are still in flight (ahead is wave-uniform; vmcnt takes an immediate)
template <int PPWv, int N>
__device__ __forceinline__ void pl_wait_steps(int ahead) {
    if constexpr (N == 0) {
        PL_WAIT_VM(0);
    } else {
        if (ahead >= N)
            PL_WAIT_VM(N * PPWv);
        else
            pl_wait_steps<PPWv, N - 1>(ahead);
    }
}

// D = how many K steps the DMA cursor runs ahead (D < R).  DBG (measurement builds, wrong results): 1 no vmcnt wait in
// front of the barrier, 2 no barrier, 4 no DMA in the K loop, 8 no epilogue, 16 no fp32 stores, 32 no store drain at the tile end
// Wave tile = NI x MI accumulator blocks of 32 x 32 (n x m): 2 x 4 at two waves per SIMD (256 registers), 4 x 4 at one
// wave per SIMD (512 registers; a third fewer fragment bytes read from LDS per MFMA -- the kernel is power-bound, so
// bytes moved per flop, not stalls, set its rate: profiles/r03_pl_ablation.txt).
// LOOP_ = 1 (r06, thin wave tiles, ONE tile per workgroup): a K loop built for a wave tile with 2-MFMA groups.  Ablation of the
// default loop on the 128 x 64 tile (profiles/r06_pl_small_tiles.txt): 930 cycles per K step, of which MFMAs + fragment reads alone
// are 570 (192 cycles of matrix issue + two exposed LDS round trips: the default loop hides them behind 8-MFMA groups), the DMA
// issue 220, waits and barrier the rest.  Here step s + 1's fragments are ALL read right behind the barrier, under step s's MFMAs
// (two register sets, loop unrolled by two), and the DMA of step s + R -- into the stage step s just vacated, so the ring runs
// R - 1 steps ahead with the same LDS -- is spread over the MFMA gaps.  Same products in the same order per accumulator.
// LOOP_ = 3 (r06, thin tiles, one tile per workgroup): the operands travel global -> REGISTERS -> LDS instead of by LDS-DMA.  Every
// restructuring of the LDS-DMA loop measured the same ~900 cycles per K step for a lone thin workgroup, whatever the ring depth: a
// wave keeps only a few global_load_lds in flight (6 pieces = 6 KiB per wave at two steps of prefetch; issuing more steps ahead did
// not raise it), so one CU fetches ~24 KiB per memory round trip (~25 - 30 GB/s) and the K loop waits for memory.  Plain
// global_load_dwordx4 into registers has the full vmcnt depth: D steps (D x PPW x 16 B per lane) are in flight per wave, written to
// a two-stage LDS ring one step ahead of their use.  Same pieces, same LDS image, same fragment reads, same MFMA order: same bits.
template <int WN_, int WM_, int R_, int D_ = 2, int DBG_ = 0, int NI_ = 2, int MI_ = 4, int LOOP_ = 0>
struct PlCfg {
    static constexpr int WN = WN_, WM = WM_, R = R_, D = D_, DBG = DBG_, NI = NI_, MI = MI_, LOOP = LOOP_;
    static_assert(LOOP_ == 0 || LOOP_ == 3 || (DBG_ == 0 && NI_ * MI_ <= 4 && R_ >= 3), "the pipelined loop is for thin wave tiles (fragment double buffer: registers)");
    static_assert(LOOP_ != 3 || (DBG_ == 0 && NI_ * MI_ <= 4), "the register-staged loop is for thin wave tiles");
    static_assert(LOOP_ != 2 || R_ >= 5, "LOOP 2 (one barrier per two steps) needs R >= 5: R - 4 steps stay in flight across a barrier");
    static constexpr int NW = WN * WM, NT = 64 * NW;
    static constexpr int TN = 32 * NI * WN, TM = 32 * MI * WM;
    static constexpr int WPS = NW > 4 ? 2 : 1;  // waves per SIMD of one workgroup
    static constexpr int SA = TM * 64, SW = TN * 64, STAGE = SA + SW;  // bytes per K step of 16
    // DMA pieces of 1 KiB per K step.  TM >= 64: one piece = (plane, k group) of a 64-row block.  TM == 32 (r06, small calls):
    // one piece = one plane of the tile's 32 rows, lanes 0..31 k group 0, lanes 32..63 k group 1 (the source address of a
    // global_load_lds is per lane, only the LDS side is contiguous) -- the same [plane][k group][TM rows][16 B] image in LDS.
    static constexpr int NPA = TM >= 64 ? 4 * (TM / 64) : 2, NPW = 4 * (TN / 64), NP = NPA + NPW;
    static constexpr int PPW = (NP + NW - 1) / NW;  // DMA pieces per wave per step; RAGGED: waves >= NP % NW issue one fewer
    static constexpr bool RAGGED = NP % NW != 0;
    static constexpr int LDS = R * STAGE;
    static constexpr bool EPI_T = NW == 8;  // fp32 epilogues go through a 4 KiB LDS block per wave (row-major global accesses)
    static_assert((TM % 64 == 0 || TM == 32) && TN % 64 == 0, "tile shape");
    static_assert((D_ - 1) * PPW < 48, "vmcnt bookkeeping: vmcnt < 64");
    static_assert(LOOP_ == 3 ? (D_ >= 2 && R_ == 2) : (D_ >= 2 && D_ < R_), "prefetch distance (LOOP 3: D register slots, two LDS stages)");
    // PL_ROWLN, thin tiles (r06): the row-major epilogue block of the 8 waves (32 KiB) lives in the ONE ring stage the DMA cursor
    // does not own during an epilogue (R - D == 1: the stage of the last K step, all of whose fragment reads are behind that
    // step's barrier) instead of behind the ring -- what lets a 4-stage ring of 34 / 36 KiB stages fit the CU's 160 KiB
    static constexpr bool SCR_IN_RING = NW == 8 && R_ - D_ == 1 && MI_ <= 2 && STAGE >= NW * 4096;
};

// The fp32 epilogue's arithmetic, in two halves with contraction off: the row factors are applied in the accumulator
// layout (lane = row), the column constants either there or after the transposition (lane = 4 fixed columns) -- every
// tile configuration rounds the same way, results do not depend on the tiling.
__device__ __forceinline__ float pl_row_part(float a, float sa, float frs, bool fold) {
#pragma clang fp contract(off)
    float t = a * sa;
    if (fold) t = frs * t;
    return no_fuse(t);
}
__device__ __forceinline__ float pl_col_part(float t, float sw, float fmu, float fg, bool unfold, float fb) {
#pragma clang fp contract(off)
    t = no_fuse(t * sw);   // every product rounded on its own in every instantiation (common.h no_fuse)
    if (unfold) t = t + no_fuse(fmu * fg);
    return t + fb;
}
__device__ __forceinline__ float pl_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// value of lane (src_bytes / 4) (ds_bpermute: a lane crossbar, no memory)
__device__ __forceinline__ float lane_bcast(float v, int src_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_bytes, __builtin_bit_cast(int, v)));
}

template <int EPI, bool SWAP, typename C>
__global__ __launch_bounds__(C::NT, (C::NI * C::MI > 8) ? 1 : 2) void gemm_pl_kernel(PlParams p) {
    constexpr int TN = C::TN, TM = C::TM, R = C::R, PPW = C::PPW, NW = C::NW, D = C::D, DBG = C::DBG;
    constexpr int NI = C::NI, MI = C::MI;
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32_ = lane & 31, hi_ = lane >> 5;
    const int wn = wave % C::WN, wm = wave / C::WN;
    const int pair = (C::WPS == 2) ? (wave >> 2) : 0;  // the two waves of a SIMD inside one workgroup: staggered DMA issue
    if ((int)blockIdx.x >= p.ntiles) return;
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nk = p.nk, kblocks = p.K >> 5;
    const int total = my_tiles * nk;
    long long t_start = 0;
    if (p.cycles && blockIdx.x == 0 && tid == 0) t_start = __builtin_amdgcn_s_memtime();
    if (p.stagger > 0) {
        // de-phase the persistent workgroups: started together they all reach their (HBM-heavy) epilogues together
        const int n_sleep = (((int)blockIdx.x >> 3) & 7) * p.stagger;
        for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(32);
    }

    // ---- DMA pieces of this wave: piece q = wave + NW * j ------------------------------------------------------
    unsigned pv_off[PPW];   // byte offset inside the tile's operand panel (+ lane * 16)
    int pl_dst[PPW];        // byte offset inside a stage
    bool p_is_w[PPW];
    // RAGGED configurations: the last piece slot exists only for the first NP % NW waves (wave-uniform)
    const bool full = !C::RAGGED || wave < C::NP % NW;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        int q = wave + NW * j;
        if (C::RAGGED && q >= C::NP) q = wave;  // never issued (guarded by `full`); keeps the arithmetic in range
        const bool is_w = q >= C::NPA;
        const int qq = is_w ? q - C::NPA : q;
        p_is_w[j] = is_w;
        pl_dst[j] = (is_w ? C::SA : 0) + qq * 1024;
        if (TM == 32 && !is_w) {  // piece = plane qq of the tile's 32 rows: lanes 0..31 k group 0, lanes 32..63 k group 1
            pv_off[j] = (unsigned)qq * 4096u + (unsigned)(lane >> 5) * 1024u + (unsigned)(lane & 31) * 16u;
        } else {
            const int rowsb = is_w ? TN / 64 : TM / 64;
            const int plkg = qq / rowsb, rb = qq % rowsb;
            pv_off[j] = (unsigned)rb * (unsigned)kblocks * 8192u + (unsigned)(plkg >> 1) * 4096u + (unsigned)(plkg & 1) * 1024u +
                        (unsigned)lane * 16u;
        }
    }
    auto piece_live = [&](int j) { return !C::RAGGED || j < PPW - 1 || full; };
    // the DMA cursor runs two steps ahead of the MFMAs, across tile boundaries
    const unsigned char *d_a = nullptr, *d_w = nullptr;  // panel bases of the cursor's tile (wave-uniform)
    int d_tile = -1, d_k = 0, d_stage = 0;
    auto d_set_tile = [&](int ti) {
        const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
        int64_t bm;
        int bn;
        tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);
            bm += p.bm0;
        const unsigned char *ab = (p.a_split_n > 0 && bn * TN >= p.a_split_n) ? p.a2 : p.a;
        int64_t arow = bm * TM;
        if (p.a_rpg > 0) arow = (arow / p.a_rpg) * p.a_gstride + p.a_goff + arow % p.a_rpg;  // whole tiles: a_rpg % TM == 0
        d_a = ab + (arow >> 6) * (int64_t)kblocks * 8192 + (TM == 32 ? (arow & 63) * 16 : 0);
        d_w = p.w + (int64_t)bn * (TN / 64) * (int64_t)kblocks * 8192;
        d_tile = ti;
        d_k = 0;
    };
    auto dma_step = [&]() {  // issue the pieces of K step d_k of tile d_tile into stage d_stage, advance the cursor
        const unsigned koff = (unsigned)(d_k >> 1) * 8192u + (unsigned)(d_k & 1) * 2048u;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (!piece_live(j)) continue;
            const unsigned char *src = (p_is_w[j] ? d_w : d_a) + koff + pv_off[j];
            __builtin_amdgcn_global_load_lds((pl_glob_t *)src, (pl_lds_t *)(pl_smem + d_stage * C::STAGE + pl_dst[j]), 16, 0, 0);
        }
        d_stage = d_stage + 1 == R ? 0 : d_stage + 1;
        if (++d_k == nk && d_tile + 1 < my_tiles) d_set_tile(d_tile + 1);
    };

    // ---- fragments ---------------------------------------------------------------------------------------------
    // A (activations, b slot unless SWAP): X = hi plane, Y = lo plane, 4 blocks of 32 rows; W: Ph / Pl, 2 blocks
    u32x4 X[MI], Y[MI], Ph[NI], Pl[NI];
    const int a_fr = (hi_ * TM + wm * (32 * MI) + r32_) * 16;
    const int w_fr = C::SA + (hi_ * TN + wn * (32 * NI) + r32_) * 16;
    auto rdA = [&](u32x4 (&f)[MI], int stage, int plane) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            f[mi] = *reinterpret_cast<const u32x4 *>(pl_smem + stage * C::STAGE + a_fr + plane * (2 * TM * 16) + mi * 512);
    };
    auto rdW = [&](u32x4 (&f)[NI], int stage, int plane) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            f[ni] = *reinterpret_cast<const u32x4 *>(pl_smem + stage * C::STAGE + w_fr + plane * (2 * TN * 16) + ni * 512);
    };
    f32x16 acc[NI][MI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;
    };
    zero_acc();
    auto mfma1 = [&](const u32x4 &wf, const u32x4 &af, f32x16 &c) {
        if constexpr (SWAP)
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, wf), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, af), c, 0, 0, 0);
    };
    // one MFMA group (NI * MI blocks) with `hook(i)` called after every second MFMA (NI * MI / 2 issue slots)
    constexpr int SLOTS = NI * MI / 2;
    auto group = [&](u32x4 (&wf)[NI], u32x4 (&af)[MI], auto &&hook) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                mfma1(wf[ni], af[mi], acc[ni][mi]);
                if ((ni * MI + mi) & 1) {  // after every second MFMA in issue order (== mi & 1 for even MI; MI = 5 needs the count)
                    __builtin_amdgcn_sched_barrier(0);
                    hook((ni * MI + mi) >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
    auto nohook = [](int) {};

    // ---- prologue ----------------------------------------------------------------------------------------------
    d_set_tile(0);
    // LOOP 3: register staging -- slot u of stg holds the pieces of step (u mod D); the cursor (d_k) is the next step to load
    u32x4 stg[C::LOOP == 3 ? D : 1][PPW];
    auto ld_step = [&](int slot) {   // (slot is a constant after unrolling)
        const unsigned koff = (unsigned)(d_k >> 1) * 8192u + (unsigned)(d_k & 1) * 2048u;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            if (piece_live(j)) stg[slot][j] = *reinterpret_cast<const u32x4 *>((p_is_w[j] ? d_w : d_a) + koff + pv_off[j]);
        ++d_k;
    };
    auto st_step = [&](int slot, int stage) {
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            if (piece_live(j)) *reinterpret_cast<u32x4 *>(pl_smem + stage * C::STAGE + pl_dst[j] + lane * 16) = stg[slot][j];
    };
    int st = 0;  // stage of the current step
    if constexpr (C::LOOP == 3) {
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (u < total) ld_step(u);
        PL_WAIT_VM(0);
        st_step(0, 0);
        if (D < total) ld_step(0);
        lds_barrier();
    } else {
        dma_step();
#pragma unroll
        for (int d = 1; d < (C::LOOP == 1 ? R : (C::LOOP == 2 ? R - 1 : D)); ++d)   // LOOP 1: every stage is filled, the cursor then runs R steps ahead (LOOP 2: R - 1)
            if (total > d) dma_step();
        PL_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
    }
    rdA(X, 0, 0);
    rdW(Pl, 0, 1);
    // LOOP == 1: second fragment set (the loop alternates between the two) and the rest of step 0's fragments
    u32x4 X2[MI], Y2[MI], Ph2[NI], Pl2[NI];
    if constexpr (C::LOOP >= 1) {
        rdW(Ph, 0, 0);
        rdA(Y, 0, 1);
    }

    for (int ti = 0; ti < my_tiles; ++ti) {
        // PL_GEGLU: the tile's per-row operand scales and the wave's 64 weight-row scales (one per lane) are requested HERE,
        // a whole K loop before the epilogue needs them (5 registers): the epilogue then starts without a memory round trip
        float pf_sa[MI], pf_sw = 0.0f;
        if constexpr (EPI == PL_GEGLU && NI == 2) {
            const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
            int64_t bm;
            int bn;
            tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);
            bm += p.bm0;
            int r32 = r32_, hi = hi_;
            asm volatile("" : "+v"(r32), "+v"(hi));
            const bool second = p.a_split_n > 0 && bn * TN >= p.a_split_n;
            const float *ascl = second ? p.a2_scale : p.a_scale;
            const float *abase = ascl ? ascl : p.w_scale;
            int64_t mmax = ascl ? p.M - 1 : 0;
            asm volatile("" : "+v"(abase), "+v"(mmax));
            const int64_t m_w0 = bm * TM + wm * (32 * MI);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int64_t m = m_w0 + mi * 32 + r32;
                pf_sa[mi] = abase[m < mmax ? m : mmax];
            }
            const int n_w0 = bn * TN + wn * (32 * NI);
            pf_sw = p.w_scale[n_w0 < p.N ? n_w0 + hi * 32 + r32 : 0];
        }
        // PL_TSCORE / PL_TPV: everything the epilogue reads from memory -- row scales and statistics, column constants, launch 2's
        // softmax weights -- is requested HERE, a whole K loop ahead (one wave per SIMD: nothing else would hide the round trips;
        // 60 registers of the 512 this configuration has).  Unconditional loads from clamped addresses, as above.
        float pt_sa[MI], pt_mu[MI], pt_rs[MI], pt_c0 = 0.0f, pt_c1 = 0.0f, pt_c2 = 0.0f, pt_vb = 1.0f;
        f32x4 pt_p[10];
        if constexpr (EPI == PL_TSCORE || EPI == PL_TPV) {
            const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
            int64_t bm;
            int bn;
            tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);
            bm += p.bm0;
            int r32 = r32_, hi = hi_;
            asm volatile("" : "+v"(r32), "+v"(hi));
            const int64_t m_w0 = bm * TM + wm * (32 * MI);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int64_t m = m_w0 + mi * 32 + r32;
                m = m < p.M ? m : p.M - 1;
                pt_sa[mi] = p.a_scale[m];
                pt_mu[mi] = p.fold_stats[2 * m];
                pt_rs[mi] = p.fold_stats[2 * m + 1];
            }
            const int n_w0 = bn * TN + wn * (32 * NI);
            const int ln = hi * 32 + r32, ncl = n_w0 < p.N ? n_w0 + ln : 0;
            pt_c0 = p.w_scale[ncl];
            const bool isq = EPI == PL_TSCORE && wn == 0;
            const float *fxp = isq ? p.fold_b : p.fold_u;
            const float *fxb = fxp ? fxp : p.w_scale;
            pt_c1 = fxb[ncl] * (fxp ? 1.0f : 0.0f);
            int seq = (int)(bm * 64) + wm * 32 + r32;
            seq = seq < p.t_nseq ? seq : 0;
            if constexpr (EPI == PL_TSCORE) {
                pt_c2 = (isq ? p.q_scale : p.k_scale)[ln];
            } else {
                const float *vbp = p.v_bound_dev ? p.v_bound_dev + (seq / p.t_seqs_per_clip) * p.v_bound_stride : p.w_scale;
                pt_vb = p.v_bound_dev ? *vbp : 1.0f + 0.0f * *vbp;
                const float *pp = p.tp + ((int64_t)seq * p.t_heads + (n_w0 < p.N ? n_w0 >> 6 : 0)) * 40;
#pragma unroll
                for (int q = 0; q < 10; ++q) pt_p[q] = *reinterpret_cast<const f32x4 *>(pp + q * 4);
            }
        }
        if constexpr (C::LOOP == 3) {
            // register-staged loop (single tile: total == nk).  At the top of step s the registers of step s + 1 have landed: they go
            // to the stage step s - 1 was read from (every wave finished with it before barrier(s - 1)) and the loads of step
            // s + 1 + D take their place; then the plain loop's three groups with its fragment timing.
            for (int k0 = 0; k0 < nk; k0 += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    const int s = k0 + u;
                    if (s < nk) {   // wave-uniform
                        const int st1 = st ^ 1;
                        if (s + 1 < total) {
                            const int ahead = total - 2 - s;   // steps s + 2 .. min(s + D, total - 1) stay in flight
                            if (full)
                                pl_wait_steps<PPW, D - 1>(ahead);
                            else
                                pl_wait_steps<(PPW > 1 ? PPW - 1 : 0), D - 1>(ahead);
                            st_step((u + 1) % D, st1);
                            if (s + 1 + D < total) ld_step((u + 1) % D);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        group(Pl, X, [&](int i) {
                            if (i == 0) {
                                rdW(Ph, st, 0);
                                rdA(Y, st, 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                        group(Ph, X, nohook);
                        lds_barrier();
                        if (s + 1 < total) {
                            rdA(X, st1, 0);
                            rdW(Pl, st1, 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        group(Ph, Y, nohook);
                        st = st1;
                    }
                }
            }
        } else if constexpr (C::LOOP >= 1) {
            // pipelined loop: [step s + 1 landed: vmcnt, barrier] [read ALL fragments of s + 1] [MFMAs of s, the DMA pieces of step
            // s + R (into the stage s came from) in their gaps]
            // LOOP 2: ONE wait + barrier per TWO steps (even steps only; a tile's step count is even or its last step syncs alone).
            // At the barrier of even step s the steps s + 1 and s + 2 have landed and every wave is done reading the stages of steps
            // <= s; step s then issues step s - 1 + R (into the stage of s - 1), step s + 1 issues s + R: the cursor runs R - 1 ahead.
            auto kstep = [&](u32x4 (&cX)[MI], u32x4 (&cY)[MI], u32x4 (&cPh)[NI], u32x4 (&cPl)[NI], u32x4 (&nX)[MI], u32x4 (&nY)[MI],
                             u32x4 (&nPh)[NI], u32x4 (&nPl)[NI], int s, bool sync) {
                const int st1 = st + 1 == R ? 0 : st + 1;
                if (C::LOOP == 1 || sync) {
                    if constexpr (C::LOOP == 1) {   // steps s + 2 .. min(s + R - 1, total - 1) may stay in flight
                        const int ahead = total - 2 - s;
                        if (full)
                            pl_wait_steps<PPW, R - 2>(ahead);
                        else
                            pl_wait_steps<(PPW > 1 ? PPW - 1 : 0), R - 2>(ahead);
                    } else {                        // steps s + 3 .. min(s - 2 + R, total - 1) may stay in flight
                        const int ahead = total - 3 - s;
                        if (full)
                            pl_wait_steps<PPW, (R > 4 ? R - 4 : 0)>(ahead);
                        else
                            pl_wait_steps<(PPW > 1 ? PPW - 1 : 0), (R > 4 ? R - 4 : 0)>(ahead);
                    }
                    lds_barrier();
                }
                if (s + 1 < total) {
                    rdA(nX, st1, 0);
                    rdW(nPl, st1, 1);
                    rdW(nPh, st1, 0);
                    rdA(nY, st1, 1);
                }
                // wave-uniform; the cursor stands on step s + R (stage == st) or, LOOP 2, s - 1 + R (the stage before st)
                const bool more = C::LOOP == 1 ? s + R < total : s - 1 + R < total;
                unsigned koff = 0;
                const unsigned char *ca = d_a, *cw = d_w;
                const int cst = d_stage;
                if (more) koff = (unsigned)(d_k >> 1) * 8192u + (unsigned)(d_k & 1) * 2048u;
                constexpr int HOOKS = 3 * SLOTS, PPH = (PPW + HOOKS - 1) / HOOKS;
                auto dma_hook = [&](int h) {
#pragma unroll
                    for (int u = 0; u < PPH; ++u) {
                        const int j = PPH * h + u;
                        if (more && j < PPW && piece_live(j)) {
                            const unsigned char *src = (p_is_w[j] ? cw : ca) + koff + pv_off[j];
                            __builtin_amdgcn_global_load_lds((pl_glob_t *)src, (pl_lds_t *)(pl_smem + cst * C::STAGE + pl_dst[j]), 16, 0, 0);
                        }
                    }
                };
                __builtin_amdgcn_sched_barrier(0);
                group(cPl, cX, [&](int i) { dma_hook(i); });               // W lo . A hi
                group(cPh, cX, [&](int i) { dma_hook(SLOTS + i); });       // W hi . A hi
                group(cPh, cY, [&](int i) { dma_hook(2 * SLOTS + i); });   // W hi . A lo
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    d_stage = d_stage + 1 == R ? 0 : d_stage + 1;
                    if (++d_k == nk && d_tile + 1 < my_tiles) d_set_tile(d_tile + 1);
                }
                st = st1;
            };
            int k = 0;
            for (; k + 1 < nk; k += 2) {
                kstep(X, Y, Ph, Pl, X2, Y2, Ph2, Pl2, ti * nk + k, true);
                kstep(X2, Y2, Ph2, Pl2, X, Y, Ph, Pl, ti * nk + k + 1, false);
            }
            if (k < nk) {   // odd step count: the next tile starts from the first set again
                kstep(X, Y, Ph, Pl, X2, Y2, Ph2, Pl2, ti * nk + k, true);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    X[mi] = X2[mi];
                    Y[mi] = Y2[mi];
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    Ph[ni] = Ph2[ni];
                    Pl[ni] = Pl2[ni];
                }
            }
        } else
        for (int k = 0; k < nk; ++k) {
            const int s = ti * nk + k;
            const bool more = (DBG & 4) ? false : s + D < total;  // wave-uniform
            const int st1 = st + 1 == R ? 0 : st + 1;
            // the cursor state is advanced by whichever group issues; pieces are spread over the MFMA gaps
            unsigned koff = 0;
            const unsigned char *ca = d_a, *cw = d_w;
            const int cst = d_stage;
            if (more) koff = (unsigned)(d_k >> 1) * 8192u + (unsigned)(d_k & 1) * 2048u;
            auto dma_piece = [&](int j) {
                if (j < PPW && piece_live(j)) {
                    const unsigned char *src = (p_is_w[j] ? cw : ca) + koff + pv_off[j];
                    __builtin_amdgcn_global_load_lds((pl_glob_t *)src, (pl_lds_t *)(pl_smem + cst * C::STAGE + pl_dst[j]), 16, 0,
                                                     0);
                }
            };
            auto dma_slot = [&](int i) {  // slot i of SLOTS: the pieces are spread over the MFMA gaps of a group
                constexpr int PPS = (PPW + SLOTS - 1) / SLOTS;  // 1 or 2 for the big tiles; thin wave tiles (r06) take more
#pragma unroll
                for (int u = 0; u < PPS; ++u) dma_piece(PPS * i + u);
            };
            // G1: W lo . A hi   | after the first MFMA pair: reads W hi, A lo of this step (their wait then sits in front
            //                     of G2 and the wait in front of G1 covers only the fragments prefetched by the last G3)
            group(Pl, X, [&](int i) {
                if (i == 0) {
                    rdW(Ph, st, 0);
                    rdA(Y, st, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more && pair == 0) dma_slot(i);
            });
            // G2: W hi . A hi
            group(Ph, X, [&](int i) {
                if (more && pair == 1) dma_slot(i);
            });
            if (more) {
                d_stage = d_stage + 1 == R ? 0 : d_stage + 1;
                if (++d_k == nk && d_tile + 1 < my_tiles) d_set_tile(d_tile + 1);
            }
            // this wave's share of step s + 1 has landed: everything but the pieces of steps s + 2 .. s + D
            // (steps s + 2 .. min(s + D, total - 1) may stay in flight: min(D - 1, total - 2 - s) of them)
            if constexpr ((DBG & 1) == 0) {
                const int ahead = total - 2 - s;
                if (full)
                    pl_wait_steps<PPW, D - 1>(ahead);
                else
                    pl_wait_steps<(PPW > 1 ? PPW - 1 : 0), D - 1>(ahead);
            }
            if constexpr ((DBG & 2) == 0) lds_barrier();
            // G3: W hi . A lo   | reads A hi, W lo of the next step
            if (s + 1 < total) {
                rdA(X, st1, 0);
                rdW(Pl, st1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            group(Ph, Y, nohook);
            st = st1;
        }
        // ---- epilogue of tile ti -------------------------------------------------------------------------------
        {
            const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
            int64_t bm;
            int bn;
            tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);
            bm += p.bm0;
            // the epilogue's lane-dependent address arithmetic must not be hoisted out of the tile loop (loop-invariant
            // code motion would keep dozens of addresses alive across the K loop: spills, and with them a vmcnt(0) in
            // front of every scratch reload): launder the lane coordinates here
            int r32 = r32_, hi = hi_;
            asm volatile("" : "+v"(r32), "+v"(hi));
            const bool second = p.a_split_n > 0 && bn * TN >= p.a_split_n;
            const float *ascl = second ? p.a2_scale : p.a_scale;
            const float ascl_c = second ? p.a2_scale_const : p.a_scale_const;
            const int n_w0 = bn * TN + wn * (32 * NI);      // first column of this wave
            const int64_t m_w0 = bm * TM + wm * (32 * MI);  // first row of this wave
            if constexpr (DBG & 8) {
                // measurement build: no epilogue (the accumulators stay live through a store that never happens)
                float t = 0.0f;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int e = 0; e < 16; ++e) t += acc[ni][mi][e];
                if (t == 12345.678f) p.c[lane] = t;
            } else if constexpr (!SWAP) {
                // lane = row m_w0 + mi * 32 + r32; run c of block ni: columns n_w0 + ni * 32 + c * 16 + hi * 8 + 0..7
                // per-row activation scales.  The load is unconditional and branch-free (a dummy address and an
                // arithmetic select when there is no per-row array): a load on one side of a branch stays "pending" on
                // the other side for the compiler's waitcnt pass, which then drains vmcnt -- the LDS-DMA ring -- in the K loop
                float sa[MI];
                {
                    const float *abase = ascl ? ascl : p.w_scale;
                    float has = ascl ? 1.0f : 0.0f;
                    int64_t mmax = ascl ? p.M - 1 : 0;
                    asm volatile("" : "+v"(abase), "+v"(has), "+v"(mmax));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int64_t m = m_w0 + mi * 32 + r32;
                        float av;
                        if constexpr (EPI == PL_GEGLU && NI == 2)
                            av = pf_sa[mi];  // requested before the K loop
                        else if constexpr (EPI == PL_TSCORE || EPI == PL_TPV)
                            av = pt_sa[mi];
                        else
                            av = abase[m < mmax ? m : mmax];
                        sa[mi] = ascl_c * (av * has + (1.0f - has));  // exact: av (has = 1) or 1 (has = 0, av finite)
                    }
                }
                // row statistics of the folded LayerNorm (unconditional loads from a valid address, see above)
                float fmu[MI], frs[MI];
                {
                    const float *sbase = p.fold_stats ? p.fold_stats : p.w_scale;
                    int64_t mmax = p.fold_stats ? p.M - 1 : 0;
                    asm volatile("" : "+v"(sbase), "+v"(mmax));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        if constexpr (EPI == PL_TSCORE || EPI == PL_TPV) {
                            fmu[mi] = pt_mu[mi];
                            frs[mi] = pt_rs[mi];
                        } else {
                            const int64_t m = m_w0 + mi * 32 + r32;
                            const float *sp = sbase + 2 * (m < mmax ? m : mmax);
                            fmu[mi] = sp[0];
                            frs[mi] = sp[p.fold_stats ? 1 : 0];
                        }
                    }
                }
                if constexpr (EPI == PL_F32 || EPI == PL_UNPATCH) {
                    float *cb = p.c;
                    int64_t ldc = p.ldc;
                    int ncol0 = n_w0;
                    if (p.c_split_n > 0 && n_w0 >= p.c_split_n) {
                        cb = p.c2;
                        ldc = p.ldc2;
                        ncol0 = n_w0 - p.c_split_n;
                    }
                    // Rows are bounded by wave-uniform buffer descriptors (loads of rows >= M return 0, stores are dropped):
                    // no divergent branch, so every load is waited for on every path -- a load left pending on a skipped
                    // path would make the compiler drain vmcnt (and with it the LDS-DMA ring) inside the K loop.
                    int64_t vr64 = p.M - m_w0;
                    const int vr = vr64 > 32 * MI ? 32 * MI : (vr64 < 0 ? 0 : (int)vr64);
                    int vc = p.N - n_w0;
                    if (vc > 32 * NI) vc = 32 * NI;
                    const auto c_rs = x3_rsrc(cb + m_w0 * ldc + ncol0, vr > 0 && vc > 0 ? ((vr - 1) * (int)ldc + vc) * 4 : 0);
                    // no residual: a zero-sized descriptor, the loads below return 0 and touch no memory
                    const auto r_rs = x3_rsrc(p.residual ? p.residual + m_w0 * p.ldr + n_w0 : cb,
                                              p.residual && vr > 0 && vc > 0 ? ((vr - 1) * (int)p.ldr + vc) * 4 : 0);
                    // Column constants: lane l holds the ones of the wave's column l (one load each per tile); a chunk's 8
                    // columns are fetched from those lanes by ds_bpermute.  With the residual reloaded block by block (the
                    // next chunk's rows are requested as soon as this chunk's are consumed) the epilogue waits for memory
                    // ONCE per tile instead of once per 8-column chunk.
                    const bool has_fold = p.fold_stats != nullptr;
                    if constexpr (C::EPI_T) {
                        // Row-major global accesses through a wave-private 4 KiB LDS block.  In the accumulator layout a
                        // lane owns 16-byte pieces of ITS row, so one store instruction touches 32 rows x 2 pieces: 32
                        // partly written 128-byte lines per instruction, and the texture path -- not HBM -- bounded the
                        // epilogue (15 us per 256 x 256 tile written, 30 us with the residual read the same way;
                        // profiles/r03_epilogue_probe.txt).  A 32 x 32 block goes through LDS instead (XOR-swizzled 16-byte
                        // chunks, no bank conflicts either way) and is stored / its residual loaded as 8 full lines per
                        // instruction: lane -> row i * 8 + lane / 8, chunk lane % 8.
                        unsigned char *scr = pl_smem + C::LDS + wave * 4096;
                        const int rrow = hi * 4 + (r32 >> 3), rch = r32 & 7;
                        const int rd0 = rrow * 128 + ((rch ^ rrow) * 16);
                        const int wr0 = r32 * 128, wsw = r32 & 7;
                        // global offsets: one lane part per leading dimension + a wave-uniform (scalar) block part
                        const int vo_r = (rrow * (int)p.ldr + rch * 4) * 4, vo_c = (rrow * (int)ldc + rch * 4) * 4;
                        auto so = [&](int blk, int i, int ld) {  // blk = ni * MI + mi
                            return (((blk % MI) * 32 + i * 8) * ld + (blk / MI) * 32) * 4;
                        };
                        u32x4 res[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) res[i] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, vo_r, so(0, i, (int)p.ldr), 0);
                        // PL_UNPATCH: wave-uniform bases.  A 32-row block = 32 consecutive patches (b, t, gy, gx0 ..) of one patch row
                        // (gw % 32 == 0), a 32-column block = 4 pixel rows (p1) x 8 pixels (p2) of one (channel, frame) plane of the
                        // patch (p == 8): only the block bases need divisions, and they are uniform
                        int64_t up_rowb[MI], up_colb[NI];
                        if constexpr (EPI == PL_UNPATCH) {
                            const int gw = p.up_W >> 3, gh = p.up_H >> 3;
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                int64_t rr = m_w0 + mi * 32;
                                rr = rr < p.M ? rr : 0;
                                const int gx0 = (int)(rr % gw); rr /= gw;
                                const int gy = (int)(rr % gh); rr /= gh;
                                const int tt = (int)(rr % p.up_t);
                                const int64_t b = rr / p.up_t;
                                up_rowb[mi] = ((b * p.up_C * p.up_F + p.up_f0 + tt * p.up_pt) * p.up_H + gy * 8) * (int64_t)p.up_W + gx0 * 8;
                            }
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const int n0 = n_w0 + ni * 32;            // = (ch pt + j) 64 + p1_0 8
                                const int cj = n0 >> 6, p1_0 = (n0 >> 3) & 7;
                                const int j = cj % p.up_pt, ch = cj / p.up_pt;
                                up_colb[ni] = (((int64_t)ch * p.up_F + j) * p.up_H + p1_0) * (int64_t)p.up_W;
                            }
                        }
                        // column constants of the row-major side: the lane's 4 columns of each 32-column block, once per tile
                        f32x4 swr[NI], fbr[NI], fgr[NI];
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const int nn = ni * 32 < vc ? n_w0 + ni * 32 + rch * 4 : 0;
                            const bool fold = has_fold && n_w0 + ni * 32 < p.fold_cols;
                            swr[ni] = *reinterpret_cast<const f32x4 *>(p.w_scale + nn);
                            fbr[ni] = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + nn) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                            if (fold && p.fold_b) {
                                const f32x4 fb2 = *reinterpret_cast<const f32x4 *>(p.fold_b + nn);
#pragma unroll
                                for (int e = 0; e < 4; ++e) fbr[ni][e] = pl_add(fbr[ni][e], fb2[e]);
                            }
                            fgr[ni] = (has_fold && !fold && p.fold_u) ? *reinterpret_cast<const f32x4 *>(p.fold_u + nn)
                                                                       : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            if (ni * 32 >= vc) break;
                            const bool fold = has_fold && n_w0 + ni * 32 < p.fold_cols;  // wave-uniform
                            const bool unfold = has_fold && !fold && p.fold_u;            // centred operand, plain column
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int ch = 0; ch < 4; ++ch) {  // piece (c, h) = 4 columns
                                    const int c = ch >> 1, h = ch & 1;
                                    f32x4 v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = pl_row_part(acc[ni][mi][c * 8 + h * 4 + e], sa[mi], frs[mi], fold);
                                    *reinterpret_cast<f32x4 *>(scr + wr0 + (((c * 4 + hi * 2 + h) ^ wsw) * 16)) = v;
                                }
                                __builtin_amdgcn_wave_barrier();
                                float fm[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // row means of the row-major side (centred operand only)
                                if (unfold) {
#pragma unroll
                                    for (int i = 0; i < 4; ++i) fm[i] = lane_bcast(fmu[mi], (i * 8 + rrow) * 4);
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const f32x4 x = *reinterpret_cast<const f32x4 *>(scr + rd0 + i * 1024);
                                    f32x4 v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float t = pl_col_part(x[e], swr[ni][e], fm[i], fgr[ni][e], unfold, fbr[ni][e]);
                                        if (p.residual) t = pl_add(t, __builtin_bit_cast(float, (unsigned)res[i][e]));
                                        v[e] = t;
                                    }
                                    if (ni * MI + mi + 1 < NI * MI)
                                        res[i] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, vo_r, so(ni * MI + mi + 1, i, (int)p.ldr), 0);
                                    if constexpr (DBG & 16) {
                                        if (v[0] == 12345.678f) p.c[lane] = v[1];  // measurement build: no stores
                                    } else if constexpr (EPI == PL_UNPATCH) {
                                        // element (m, n) of the token x feature result is pixel  up_rowb[mi] + (i 8 + rrow) 8  +  up_colb[ni]
                                        // + (rch / 2) W + (rch % 2) 4: the lane's 4 columns are 16 contiguous bytes, the 8 lanes of a row cover 4
                                        // pixel rows of the patch and neighbouring rows (gx, gx + 1) continue them (256-byte runs)
                                        const int64_t m = m_w0 + mi * 32 + i * 8 + rrow;
                                        if (m < p.M && n_w0 + ni * 32 < p.N)
                                            *reinterpret_cast<f32x4 *>(p.c + up_rowb[mi] + up_colb[ni] + (i * 8 + rrow) * 8 + (rch >> 1) * p.up_W +
                                                                       (rch & 1) * 4) = v;
                                    } else
                                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rs, vo_c,
                                                                               so(ni * MI + mi, i, (int)ldc), 0);
                                }
                                __builtin_amdgcn_wave_barrier();
                            }
                        }
                    } else {
                        int ncl = n_w0 + lane;
                        const bool foldl = has_fold && ncl < p.fold_cols;
                        ncl = ncl < p.N ? ncl : p.N - 1;
                        const float swl = p.w_scale[ncl];
                        float fbl = p.bias ? p.bias[ncl] : 0.0f;
                        if (foldl && p.fold_b) fbl += p.fold_b[ncl];
                        const float fgl = (has_fold && !foldl && p.fold_u) ? p.fold_u[ncl] : 0.0f;
                        auto res_off = [&](int q, int mi, int h) {
                            return (((mi * 32 + r32) * (int)p.ldr + (q >> 1) * 32 + hi * 8) + (q & 1) * 16 + h * 4) * 4;
                        };
                        u32x4 res[MI][2];
    #pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
    #pragma unroll
                            for (int h = 0; h < 2; ++h) res[mi][h] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, res_off(0, mi, h), 0, 0);
    #pragma unroll
                        for (int q = 0; q < 2 * NI; ++q) {
                            const int ni = q >> 1, c = q & 1;
                            if (ni * 32 >= vc) break;
                            const bool fold = has_fold && n_w0 + ni * 32 < p.fold_cols;  // wave-uniform
                            const bool unfold = has_fold && !fold && p.fold_u;            // centred operand, plain column
                            __builtin_amdgcn_sched_barrier(0);
                            float sw[8], fg[8], fbv[8];
    #pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int src = (hi * 8 + ni * 32 + c * 16 + j) * 4;
                                sw[j] = lane_bcast(swl, src);
                                fbv[j] = lane_bcast(fbl, src);
                                fg[j] = lane_bcast(fgl, src);
                            }
    #pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                const int c_off = ((mi * 32 + r32) * (int)ldc + ni * 32 + hi * 8) * 4;
    #pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    f32x4 v;
    #pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float t = pl_row_part(acc[ni][mi][c * 8 + h * 4 + e], sa[mi], frs[mi], fold);
                                        t = pl_col_part(t, sw[h * 4 + e], fmu[mi], fg[h * 4 + e], unfold, fbv[h * 4 + e]);
                                        if (p.residual) t = pl_add(t, __builtin_bit_cast(float, (unsigned)res[mi][h][e]));
                                        v[e] = t;
                                    }
                                    if (q + 1 < 2 * NI)
                                        res[mi][h] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, res_off(q + 1, mi, h), 0, 0);
                                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rs,
                                                                           c_off + (c * 16 + h * 4) * 4, 0, 0);
                                }
                            }
                        }
                    }
                } else if constexpr (EPI == PL_GEGLU) {
                    // blocks 2 q / 2 q + 1 = value / gate columns of the same 32 hidden columns
                    static_assert(NI % 2 == 0, "GEGLU pairs");
#pragma unroll
                    for (int q = 0; q < NI / 2; ++q) {
                        const int n_q0 = n_w0 + q * 64;
                        if (n_q0 >= p.N) break;
                        const int jb = n_q0 >> 6;  // hidden k block (32 hidden columns per 64 packed ones)
                        f32x4 sv[2][2], sg[2][2];
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int n = n_q0 + c * 16 + hi * 8 + h * 4;
                                if constexpr (NI == 2) {  // lane l holds the scale of column n_w0 + l (value 0..31, gate 32..63)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const int src = (c * 16 + hi * 8 + h * 4 + e) * 4;
                                        sv[c][h][e] = lane_bcast(pf_sw, src);
                                        sg[c][h][e] = lane_bcast(pf_sw, src + 128);
                                    }
                                } else {
                                    sv[c][h] = *reinterpret_cast<const f32x4 *>(p.w_scale + n);
                                    sg[c][h] = *reinterpret_cast<const f32x4 *>(p.w_scale + n + 32);
                                }
                            }
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            __builtin_amdgcn_sched_barrier(0);
                            const int64_t m = m_w0 + mi * 32 + r32;  // rows beyond M: the plane buffers are padded to the tile
                            unsigned char *blk = p.cp + ((m >> 6) * p.cp_kblocks + jb) * 8192 + (m & 63) * 16;
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                f32x4 va, vb;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float v0 = acc[2 * q][mi][c * 8 + e] * (sa[mi] * sv[c][0][e]);
                                    const float g0 = acc[2 * q + 1][mi][c * 8 + e] * (sa[mi] * sg[c][0][e]);
                                    const float v1 = acc[2 * q][mi][c * 8 + 4 + e] * (sa[mi] * sv[c][1][e]);
                                    const float g1 = acc[2 * q + 1][mi][c * 8 + 4 + e] * (sa[mi] * sg[c][1][e]);
                                    va[e] = (gelu_erf(g0) * v0) * p.cp_scale;
                                    vb[e] = (gelu_erf(g1) * v1) * p.cp_scale;
                                }
                                const f16x4 ha = __builtin_convertvector(va, f16x4), hb = __builtin_convertvector(vb, f16x4);
                                const f16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), f16x4);
                                const f16x4 lb = __builtin_convertvector(vb - __builtin_convertvector(hb, f32x4), f16x4);
                                unsigned char *dst = blk + (c * 2 + hi) * 1024;
                                *reinterpret_cast<u32x4 *>(dst) =
                                    __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
                                *reinterpret_cast<u32x4 *>(dst + 4096) =
                                    __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
                            }
                        }
                    }
                } else if constexpr (EPI == PL_ROWLN) {
                    // Full-row tiles (TN == N == 32 * NI * WN): out = acc (+ bias) + residual is stored as fp32 AND
                    // LayerNorm(out) (two-pass statistics like norm.hip::row_stats) is written as fp16 planes, the A
                    // operand of the GEMM that follows (reference attention.py:666-680: x = attn(x) + x; x = ff(x) + x
                    // with ff = Sequential(LayerNorm, ...)).  A lane owns 32 * NI of the row's columns; the row sums
                    // go through a [WN][TM] LDS table behind the ring.
                    static_assert(C::WM == 1 || EPI != PL_ROWLN, "row statistics: one wave row per tile");
                    float *red0 = reinterpret_cast<float *>(pl_smem + C::LDS);
                    float *red1 = red0 + C::WN * TM;
                    int64_t vr64 = p.M - m_w0;
                    const int vr = vr64 > 32 * MI ? 32 * MI : (vr64 < 0 ? 0 : (int)vr64);
                    const auto c_rs = x3_rsrc(p.c + m_w0 * p.ldc + n_w0, vr > 0 ? ((vr - 1) * (int)p.ldc + 32 * NI) * 4 : 0);
                    const auto r_rs = x3_rsrc(p.residual ? p.residual + m_w0 * p.ldr + n_w0 : p.c,
                                              p.residual && vr > 0 ? ((vr - 1) * (int)p.ldr + 32 * NI) * 4 : 0);
                    float rsum[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) rsum[mi] = 0.0f;
                    // LayerNorm gamma / beta by lane (column n_w0 + lane; a chunk's columns are fetched by ds_bpermute)
                    const int ncl = n_w0 + lane;
                    const float gml = p.ln_gamma[ncl];
                    const float btl = p.ln_beta ? p.ln_beta[ncl] : 0.0f;
                    static_assert(C::EPI_T || EPI != PL_ROWLN, "row-major epilogue block");
                    // residual in and fp32 out both go row-major through the wave's LDS block (see PL_F32): the residual is
                    // loaded as full lines, re-read in the accumulator layout, and the sum (kept in the accumulators for
                    // the statistics) takes the same way back out
                    unsigned char *scr = C::SCR_IN_RING ? pl_smem + d_stage * C::STAGE + wave * 4096
                                                        : pl_smem + C::LDS + 2 * C::WN * TM * 4 + wave * 4096;
                    const int rrow = hi * 4 + (r32 >> 3), rch = r32 & 7;
                    const int rd0 = rrow * 128 + ((rch ^ rrow) * 16);
                    const int wr0 = r32 * 128, wsw = r32 & 7;
                    const int vo_r = (rrow * (int)p.ldr + rch * 4) * 4, vo_c = (rrow * (int)p.ldc + rch * 4) * 4;
                    auto so = [&](int blk, int i, int ld) {  // blk = ni * MI + mi; wave-uniform (scalar) part of the offset
                        return (((blk % MI) * 32 + i * 8) * ld + (blk / MI) * 32) * 4;
                    };
                    u32x4 res[4];  // zero-sized descriptor without a residual: the loads return 0
#pragma unroll
                    for (int i = 0; i < 4; ++i) res[i] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, vo_r, so(0, i, (int)p.ldr), 0);
                    f32x4 swr[NI], fbr[NI];  // weight scale / bias of the lane's 4 columns per block on the row-major side
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int nn = n_w0 + ni * 32 + rch * 4;
                        swr[ni] = *reinterpret_cast<const f32x4 *>(p.w_scale + nn);
                        fbr[ni] = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + nn) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            __builtin_amdgcn_sched_barrier(0);
                            // (a) acc * row scale, accumulator layout -> LDS
#pragma unroll
                            for (int ch = 0; ch < 4; ++ch) {
                                const int c = ch >> 1, h = ch & 1;
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = pl_row_part(acc[ni][mi][c * 8 + h * 4 + e], sa[mi], 1.0f, false);
                                *reinterpret_cast<f32x4 *>(scr + wr0 + (((c * 4 + hi * 2 + h) ^ wsw) * 16)) = v;
                            }
                            __builtin_amdgcn_wave_barrier();
                            // (b) row-major: column constants, residual, fp32 store; the sum goes back into the block
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const f32x4 x = *reinterpret_cast<const f32x4 *>(scr + rd0 + i * 1024);
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    v[e] = pl_add(pl_col_part(x[e], swr[ni][e], 0.0f, 0.0f, false, fbr[ni][e]),
                                                  __builtin_bit_cast(float, (unsigned)res[i][e]));
                                if (ni * MI + mi + 1 < NI * MI)
                                    res[i] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, vo_r, so(ni * MI + mi + 1, i, (int)p.ldr), 0);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), c_rs, vo_c,
                                                                       so(ni * MI + mi, i, (int)p.ldc), 0);
                                *reinterpret_cast<f32x4 *>(scr + rd0 + i * 1024) = v;
                            }
                            __builtin_amdgcn_wave_barrier();
                            // (c) back in the accumulator layout for the statistics and the LayerNorm planes
#pragma unroll
                            for (int ch = 0; ch < 4; ++ch) {
                                const int c = ch >> 1, h = ch & 1;
                                const f32x4 v = *reinterpret_cast<const f32x4 *>(scr + wr0 + (((c * 4 + hi * 2 + h) ^ wsw) * 16));
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[ni][mi][c * 8 + h * 4 + e] = v[e];
                                rsum[mi] += (v[0] + v[1]) + (v[2] + v[3]);
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    // mean
                    float mean[MI], rstd[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const float t = rsum[mi] + swap32(rsum[mi]);
                        if (hi == 0) red0[wn * TM + mi * 32 + r32] = t;
                    }
                    lds_barrier();
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        float t = 0.0f;
#pragma unroll
                        for (int w8 = 0; w8 < C::WN; ++w8) t += red0[w8 * TM + mi * 32 + r32];
                        mean[mi] = t / (float)(32 * NI * C::WN);
                    }
                    // centred sum of squares
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        float q = 0.0f;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; r += 4) {
                                const float a = acc[ni][mi][r] - mean[mi], b = acc[ni][mi][r + 1] - mean[mi];
                                const float c2 = acc[ni][mi][r + 2] - mean[mi], d = acc[ni][mi][r + 3] - mean[mi];
                                q += (a * a + b * b) + (c2 * c2 + d * d);
                            }
                        q += swap32(q);
                        if (hi == 0) red1[wn * TM + mi * 32 + r32] = q;
                    }
                    lds_barrier();
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        float t = 0.0f;
#pragma unroll
                        for (int w8 = 0; w8 < C::WN; ++w8) t += red1[w8 * TM + mi * 32 + r32];
                        rstd[mi] = 1.0f / sqrtf(t / (float)(32 * NI * C::WN) + p.ln_eps);
                    }
                    // LayerNorm -> planes
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int kb = (n_w0 >> 5) + ni;
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float g0[4], g1[4], b0[4], b1[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int src = (hi * 8 + ni * 32 + c * 16 + e) * 4;
                                g0[e] = lane_bcast(gml, src);
                                g1[e] = lane_bcast(gml, src + 16);
                                b0[e] = lane_bcast(btl, src);
                                b1[e] = lane_bcast(btl, src + 16);
                            }
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                const int64_t m = m_w0 + mi * 32 + r32;
                                f32x4 va, vb;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    va[e] = ((acc[ni][mi][c * 8 + e] - mean[mi]) * rstd[mi] * g0[e] + b0[e]) * p.cp_scale;
                                    vb[e] = ((acc[ni][mi][c * 8 + 4 + e] - mean[mi]) * rstd[mi] * g1[e] + b1[e]) * p.cp_scale;
                                }
                                const f16x4 ha = __builtin_convertvector(va, f16x4), hb = __builtin_convertvector(vb, f16x4);
                                const f16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), f16x4);
                                const f16x4 lb = __builtin_convertvector(vb - __builtin_convertvector(hb, f32x4), f16x4);
                                unsigned char *dst = p.cp + ((m >> 6) * p.cp_kblocks + kb) * 8192 + (c * 2 + hi) * 1024 + (m & 63) * 16;
                                *reinterpret_cast<u32x4 *>(dst) =
                                    __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
                                *reinterpret_cast<u32x4 *>(dst + 4096) =
                                    __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
                            }
                        }
                    }
                } else if constexpr (EPI == PL_TSCORE) {
                    // Tile = ONE head: 128 columns [q_h | k_h]; wave (wn, wm): wn = 0 holds q_h, wn = 1 holds k_h (64 channels = blocks
                    // ni 0, 1) of the 32 sequences of half wm, block mi = time step mi.  A lane holds 32 of the 64 channels of ITS
                    // sequence for every step, its partner (hi ^ 1) the other 32.  The k wave hands k_s to the q wave of the same half
                    // through LDS, one time step per round (8 KiB per wave, double-buffered in the two ring stages the DMA cursor
                    // does not own during the epilogue, 16-byte chunks XOR-swizzled by the sequence: conflict-free both ways); a score
                    // is then a lane-local dot over 32 channels + one exchange with the partner.  Arithmetic of attn_temporal_reg
                    // (l2norm with F.normalize's eps, learned scales, SDPA scale on q, causal mask, ALiBi, exp(s - max)); the
                    // un-normalised weights e and 1 / sum e go out: 160 bytes per (sequence, head) instead of 2 x 1280 bytes of q and k.
                    static_assert(EPI != PL_TSCORE || (NI == 2 && MI == 5 && C::WN == 2 && R - D >= 2 && C::STAGE >= 24576),
                                  "q_h | k_h waves x 5 time steps; exchange buffers in the free ring stages");
                    {
                        const bool isq = wn == 0;  // wave-uniform
                        const int head = n_w0 >> 7;
                        const int ln = hi * 32 + r32;
                        const float swl = pt_c0, fxl = pt_c1, svl = pt_c2;   // requested before the K loop
                        float ss[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) ss[mi] = 0.0f;
                        // phase A: projected values (operand / weight scales, folded LayerNorm for q, mean add-back for k)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                __builtin_amdgcn_sched_barrier(0);
                                float sw[8], fx[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const int src = (ni * 32 + c * 16 + hi * 8 + e) * 4;
                                    sw[e] = lane_bcast(swl, src);
                                    fx[e] = lane_bcast(fxl, src);
                                }
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi) {
                                    const float ra = isq ? frs[mi] : 1.0f, rb = isq ? 1.0f : fmu[mi];
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        f32x4 v;
#pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                            v[e] = ra * (acc[ni][mi][c * 8 + h * 4 + e] * (sa[mi] * sw[h * 4 + e])) + rb * fx[h * 4 + e];
                                        ss[mi] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
                                        for (int e = 0; e < 4; ++e) acc[ni][mi][c * 8 + h * 4 + e] = v[e];
                                    }
                                }
                            }
                        // phase B: l2norm, learned scales, SDPA scale on q
                        float inv[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) inv[mi] = 1.0f / fmaxf(sqrtf(ss[mi] + swap32(ss[mi])), 1e-12f);
                        const float mul = isq ? p.t_scale : 1.0f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                __builtin_amdgcn_sched_barrier(0);
                                float s8[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) s8[e] = lane_bcast(svl, (ni * 32 + c * 16 + hi * 8 + e) * 4);
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                    for (int e = 0; e < 8; ++e) acc[ni][mi][c * 8 + e] = acc[ni][mi][c * 8 + e] * inv[mi] * s8[e] * mul;
                            }
                        // exchange + scores: the k wave of a half hands its k to the q wave of the same half in two rounds (steps
                        // 0..2, then 3..4: 24 / 16 KiB in the half's own free ring stage); the q wave forms d[t][s] for t >= s
                        unsigned char *xb = pl_smem + (wm ? (d_stage + 1 == R ? 0 : d_stage + 1) : d_stage) * C::STAGE;
                        const int rowb = r32 * 256, sw15 = r32 & 15;
                        float sc[15];  // d[t][s] at t (t + 1) / 2 + s
#pragma unroll
                        for (int rd = 0; rd < 2; ++rd) {
                            const int s0 = rd ? 3 : 0, s1 = rd ? 5 : 3;
                            if (!isq) {
#pragma unroll
                                for (int s = s0; s < s1; ++s)
#pragma unroll
                                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                                        for (int c = 0; c < 2; ++c)
#pragma unroll
                                            for (int h = 0; h < 2; ++h) {
                                                const int chunk = ni * 8 + c * 4 + hi * 2 + h;
                                                *reinterpret_cast<f32x4 *>(xb + (s - s0) * 8192 + rowb + ((chunk ^ sw15) << 4)) =
                                                    f32x4{acc[ni][s][c * 8 + h * 4], acc[ni][s][c * 8 + h * 4 + 1], acc[ni][s][c * 8 + h * 4 + 2],
                                                          acc[ni][s][c * 8 + h * 4 + 3]};
                                            }
                            }
                            lds_barrier();
                            if (isq) {
#pragma unroll
                                for (int s = s0; s < s1; ++s) {
                                    f32x4 kk[8];
#pragma unroll
                                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                                        for (int c = 0; c < 2; ++c)
#pragma unroll
                                            for (int h = 0; h < 2; ++h) {
                                                const int chunk = ni * 8 + c * 4 + hi * 2 + h;
                                                kk[ni * 4 + c * 2 + h] =
                                                    *reinterpret_cast<const f32x4 *>(xb + (s - s0) * 8192 + rowb + ((chunk ^ sw15) << 4));
                                            }
#pragma unroll
                                    for (int t = s; t < 5; ++t) {
                                        float d = 0.0f;
#pragma unroll
                                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                                            for (int c = 0; c < 2; ++c)
#pragma unroll
                                                for (int h = 0; h < 2; ++h)
#pragma unroll
                                                    for (int e = 0; e < 4; ++e)
                                                        d = fmaf(acc[ni][t][c * 8 + h * 4 + e], kk[ni * 4 + c * 2 + h][e], d);
                                        sc[t * (t + 1) / 2 + s] = d + swap32(d);
                                    }
                                }
                            }
                            if (rd == 0) lds_barrier();  // round 0 is read before round 1 overwrites the block
                        }
                        lds_barrier();  // every q wave is done with the buffers before the ring takes the stages back
                        if (isq && n_w0 < p.N) {
                            const float slope = p.t_alibi ? p.t_alibi[head] : 0.0f;
                            const int seq = (int)(bm * 64) + wm * 32 + r32;
                            float *pp = p.tp + ((int64_t)(seq < p.t_nseq ? seq : 0) * p.t_heads + head) * 40;
                            const bool wr = hi == 0 && seq < p.t_nseq;
#pragma unroll
                            for (int i = 0; i < 5; ++i) {
                                float mx = -INFINITY, dd[5];
#pragma unroll
                                for (int j = 0; j <= i; ++j) {
                                    dd[j] = sc[i * (i + 1) / 2 + j] - slope * (float)(i - j);
                                    mx = fmaxf(mx, dd[j]);
                                }
                                float l = 0.0f, ev[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                                for (int j = 0; j <= i; ++j) {
                                    ev[j] = expf(dd[j] - mx);
                                    l += ev[j];
                                }
                                if (wr) {
                                    *reinterpret_cast<f32x4 *>(pp + i * 8) = f32x4{ev[0], ev[1], ev[2], ev[3]};
                                    *reinterpret_cast<f32x4 *>(pp + i * 8 + 4) = f32x4{ev[4], 1.0f / l, 0.0f, 0.0f};
                                }
                            }
                        }
                    }
                } else if constexpr (EPI == PL_TPV) {
                    // Tile = V of two heads, wave (wn, wm) = v of head 2 bn + wn for the 32 sequences of half wm; block mi = time step.
                    // o_i = (sum_{j<=i} e_ij v_j) (1 / l_i), formed in place from the last step down (o_i needs v_j, j <= i, only),
                    // scaled and split into the out-projection's operand planes at the TOKEN row seq * 5 + i.
                    static_assert(EPI != PL_TPV || (NI == 2 && MI == 5), "v_h columns x 5 time steps per wave");
                    if (n_w0 < p.N) {
                        const int head = n_w0 >> 6;
                        const int ln = hi * 32 + r32;
                        const float swl = pt_c0, ful = pt_c1;   // requested before the K loop, like the softmax weights pt_p
                        const int seq = (int)(bm * 64) + wm * 32 + r32;
                        const bool live = seq < p.t_nseq;
                        const int seqc = live ? seq : 0;
                        const float so = h2_scale_of_bound(p.v_bound * pt_vb);
                        // phase A: v = xc . Wv^T (scales) + mean u
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                __builtin_amdgcn_sched_barrier(0);
                                float sw[8], fx[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const int src = (ni * 32 + c * 16 + hi * 8 + e) * 4;
                                    sw[e] = lane_bcast(swl, src);
                                    fx[e] = lane_bcast(ful, src);
                                }
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                    for (int e = 0; e < 8; ++e)
                                        acc[ni][mi][c * 8 + e] = acc[ni][mi][c * 8 + e] * (sa[mi] * sw[e]) + fmu[mi] * fx[e];
                            }
                        // phase B: P . V in place, last step first
#pragma unroll
                        for (int i = 4; i >= 0; --i) {
                            __builtin_amdgcn_sched_barrier(0);
                            const f32x4 pa = pt_p[2 * i], pb = pt_p[2 * i + 1];
                            const float e0 = pa[0], e1 = pa[1], e2 = pa[2], e3 = pa[3], e4 = pb[0];
                            const float fin = pb[1] * so;   // (1 / l_i) and the plane scale (a power of two: exact)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    float o = acc[ni][0][r] * e0;
                                    if (i >= 1) o = fmaf(acc[ni][1][r], e1, o);
                                    if (i >= 2) o = fmaf(acc[ni][2][r], e2, o);
                                    if (i >= 3) o = fmaf(acc[ni][3][r], e3, o);
                                    if (i >= 4) o = fmaf(acc[ni][4][r], e4, o);
                                    acc[ni][i][r] = o * fin;
                                }
                        }
                        // Planes of the out-projection's operand in token order.  The wave's 32 sequences x 5 steps are 160
                        // CONSECUTIVE token rows, but a lane holds rows 5 apart: stored straight from the accumulators every
                        // instruction touched 64 partly written lines (measured: 0.51 ms for this launch against 0.25 expected).
                        // Each (block, run) therefore goes through a wave-private 10 KiB LDS block -- two waves per free ring
                        // stage -- written by (sequence, step) and read back by token row: 1 KiB per store instruction.
                        unsigned char *stg = pl_smem + ((wave >> 1) ? (d_stage + 1 == R ? 0 : d_stage + 1) : d_stage) * C::STAGE + (wave & 1) * 10240;
                        const int64_t row0 = ((int64_t)(bm * 64) + wm * 32) * 5;   // first token row of this wave
                        const int lane64 = hi * 32 + r32;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int i = 0; i < 5; ++i) {
                                    f32x4 va, vb2;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        va[e] = acc[ni][i][c * 8 + e];
                                        vb2[e] = acc[ni][i][c * 8 + 4 + e];
                                    }
                                    const f16x4 ha = __builtin_convertvector(va, f16x4), hb = __builtin_convertvector(vb2, f16x4);
                                    const f16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), f16x4);
                                    const f16x4 lb = __builtin_convertvector(vb2 - __builtin_convertvector(hb, f32x4), f16x4);
                                    unsigned char *d = stg + (hi * 160 + r32 * 5 + i) * 16;   // [plane][hi][160 rows][16 B]
                                    *reinterpret_cast<u32x4 *>(d) = __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
                                    *reinterpret_cast<u32x4 *>(d + 5120) =
                                        __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
                                }
                                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes
                                __builtin_amdgcn_wave_barrier();
                                const int kb = (n_w0 >> 5) + ni;
#pragma unroll
                                for (int pl2 = 0; pl2 < 2; ++pl2)
#pragma unroll
                                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                                        for (int j = 0; j < 3; ++j) {
                                            const int rr = j * 64 + lane64;           // row inside the wave's 160
                                            const u32x4 w = *reinterpret_cast<const u32x4 *>(stg + ((pl2 * 2 + hh) * 160 + (rr < 160 ? rr : 0)) * 16);
                                            const int64_t m = row0 + rr;
                                            if (rr < 160 && m < (int64_t)p.t_nseq * 5)
                                                *reinterpret_cast<u32x4 *>(p.cp + ((m >> 6) * (int64_t)p.cp_kblocks + kb) * 8192 + pl2 * 4096 +
                                                                           (c * 2 + hh) * 1024 + (m & 63) * 16) = w;
                                        }
                                __builtin_amdgcn_wave_barrier();
                            }
                        if (live && n_w0 == 0 && hi == 0) {
#pragma unroll
                            for (int i = 0; i < 5; ++i) p.t_out_scale[(int64_t)seqc * 5 + i] = 1.0f / so;
                        }
                    }
                    lds_barrier();  // every wave is done with its staging block before the ring takes the stages back
                } else if constexpr (EPI == PL_QKPACK) {
                    // One head of Q or K per wave (64 columns = blocks 0 .. 1 of the wave tile; NI == 2): a lane holds 32 of
                    // the 64 channels of token m (block ni, run c: d = 32 ni + 16 c + 8 hi + j), its partner lane (hi ^ 1)
                    // the other 32.  RoPE pairs (2 p, 2 p + 1) are adjacent registers, the l2 norm is a lane-local sum plus
                    // one exchange with the partner, and run (ni, c) IS the 16-byte chunk (k-step 2 ni + c, half hi) of the
                    // packed attention operand: no cross-lane traffic, no fp32 Q / K in memory, no attn_pack pass.
                    static_assert(NI == 2 || EPI != PL_QKPACK, "one head (64 columns) per wave");
                    const bool is_k = n_w0 >= p.qk_k0;                     // wave-uniform
                    const int head = ((is_k ? n_w0 - p.qk_k0 : n_w0) >> 6);
                    unsigned char *outp = is_k ? p.kp : p.qp;
                    const float *svec = is_k ? p.k_scale : p.q_scale;
                    const float mul = is_k ? 1.0f : p.q_mul, so = is_k ? p.sk : p.sq;
                    const bool fold = p.fold_stats && n_w0 < p.fold_cols;
                    const bool unfold = p.fold_stats && !fold && p.fold_u;
                    const int nblk = p.qk_ntok >> 5;
                    // RoPE factors of the tile's TM rows, staged ONCE per workgroup in the two ring stages that hold no
                    // in-flight data at this point (the DMA cursor is D steps ahead: stages st .. st + D - 1 belong to the
                    // next tile, d_stage and d_stage + 1 are free until its K loop resumes): [row][32 pairs] cos in one,
                    // sin in the other, 16-byte chunks XOR-swizzled by the row so that the 32 lanes of a block read without
                    // bank conflicts.  Before: every lane fetched its own 8-byte pieces from the [n_tokens][32] tables in
                    // global memory, 128 uncoalesced loads per lane and tile, four waves the same rows -- 0.13 ms of the
                    // 0.66 ms launch.  All waves take part (and meet at the two barriers) whether their columns exist or not.
                    static_assert(EPI != PL_QKPACK || (C::STAGE >= TM * 128 && (C::LOOP == 3 ? R >= 2 : R - D >= 2) && TM % (8 * NW) == 0),
                                  "RoPE rows of a tile fit the two free ring stages");
                    // per-column constants by lane (column n_w0 + lane of the wave's 64), requested before the RoPE rows so that
                    // one memory round trip per tile covers both; chunks fetch theirs with ds_bpermute in one batch
                    float swl, fxl = 0.0f, svl;
                    {
                        const int ln = hi * 32 + r32, ncl = n_w0 < p.N ? n_w0 + ln : 0;
                        swl = p.w_scale[ncl];
                        if (fold && p.fold_b) fxl = p.fold_b[ncl];
                        if (unfold) fxl = p.fold_u[ncl];
                        svl = svec ? svec[ln] : 1.0f;
                    }
                    const bool rope = p.cosT != nullptr;  // kernel-uniform
                    unsigned char *rope_c = pl_smem + d_stage * C::STAGE;
                    unsigned char *rope_s = pl_smem + (d_stage + 1 == R ? 0 : d_stage + 1) * C::STAGE;
                    if (rope) {
                        const int ln = hi * 32 + r32;
#pragma unroll
                        for (int i = 0; i < TM / NW / 8; ++i) {
                            const int row0 = wave * (TM / NW) + i * 8;  // wave-uniform; this lane: row0 + ln / 8, position ln % 8
                            const int64_t m = bm * TM + row0 + (ln >> 3);
                            const int64_t mc = m < p.M ? m : p.M - 1;
                            const int tokr = (int)(mc % p.qk_ntok);
                            const int off = (tokr * 32 + (((ln & 7) ^ ((ln >> 3) & 7)) << 2)) * 4;
                            __builtin_amdgcn_global_load_lds((pl_glob_t *)(reinterpret_cast<const unsigned char *>(p.cosT) + off),
                                                             (pl_lds_t *)(rope_c + row0 * 128), 16, 0, 0);
                            __builtin_amdgcn_global_load_lds((pl_glob_t *)(reinterpret_cast<const unsigned char *>(p.sinT) + off),
                                                             (pl_lds_t *)(rope_s + row0 * 128), 16, 0, 0);
                        }
                        PL_WAIT_VM(0);
                        lds_barrier();
                    }
                    // rows of this lane: token index inside its sequence and the 32-token block it belongs to
                    int tok[MI];
                    int64_t seqs[MI];
                    float ss[MI];
                    if (n_w0 < p.N) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const int64_t m = m_w0 + mi * 32 + r32;
                            const int64_t mc = m < p.M ? m : p.M - 1;
                            seqs[mi] = mc / p.qk_ntok;
                            tok[mi] = (int)(mc - seqs[mi] * p.qk_ntok);
                        }
                        // phase A: projected values (scales, folded LayerNorm / mean add-back, RoPE) written back INTO the
                        // accumulators, sums of squares per row.  Column constants are loaded once per 8 columns and reused
                        // for the MI row blocks.
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) ss[mi] = 0.0f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                __builtin_amdgcn_sched_barrier(0);
                                const int rpos = (wm * (32 * MI) + r32) * 128 + (((ni * 4 + c * 2 + hi) ^ (r32 & 7)) << 4);
                                // the chunk's 8 columns of the per-column constants, from the lanes that hold them
                                float sw[2][4], fx[2][4];
#pragma unroll
                                for (int h = 0; h < 2; ++h)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const int src = (ni * 32 + c * 16 + hi * 8 + h * 4 + e) * 4;
                                        sw[h][e] = lane_bcast(swl, src);
                                        fx[h][e] = lane_bcast(fxl, src);
                                    }
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi) {
                                    const float ra = fold ? frs[mi] : 1.0f, rb = fold ? 1.0f : (unfold ? fmu[mi] : 0.0f);
#pragma unroll
                                    for (int h = 0; h < 2; ++h) {
                                        f32x4 v;
#pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                            v[e] = ra * (acc[ni][mi][c * 8 + h * 4 + e] * (sa[mi] * sw[h][e])) + rb * fx[h][e];
                                        if (rope) {  // (a + ib)(c + is) on the pairs (d, d + 1), reference attention.py:65-69
                                            const float2 cs = *reinterpret_cast<const float2 *>(rope_c + rpos + mi * 4096 + h * 8);
                                            const float2 sn = *reinterpret_cast<const float2 *>(rope_s + rpos + mi * 4096 + h * 8);
                                            const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
                                            v[0] = a0 * cs.x - b0 * sn.x; v[1] = a0 * sn.x + b0 * cs.x;
                                            v[2] = a1 * cs.y - b1 * sn.y; v[3] = a1 * sn.y + b1 * cs.y;
                                        }
                                        ss[mi] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
                                        for (int e = 0; e < 4; ++e) acc[ni][mi][c * 8 + h * 4 + e] = v[e];
                                    }
                                }
                            }
                    }
                    if (rope) lds_barrier();  // every wave is done with the staged rows before the ring takes the stages back
                    if (n_w0 < p.N) {
                        unsigned char *blkp[MI];
                        bool live[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            live[mi] = m_w0 + mi * 32 + r32 < p.M;
                            blkp[mi] = outp + ((seqs[mi] * p.qk_heads + head) * nblk + (tok[mi] >> 5)) * 8192 + hi * 512 + (tok[mi] & 31) * 16;
                        }
                        float inv[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const float t = ss[mi] + swap32(ss[mi]);
                            inv[mi] = mul * so / fmaxf(sqrtf(t), 1e-12f);  // F.normalize eps (attention.py:24-25), SDPA scale,
                            // power-of-two operand scale.  Without scale vectors (window attention, attention.py:272-275: q * scale and
                            // k as projected) there is no normalisation either
                            if (!svec) inv[mi] = mul * so;
                        }
                        // phase B: l2norm, learned scale, split, store
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                __builtin_amdgcn_sched_barrier(0);
                                float s0[4], s1[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int src = (ni * 32 + c * 16 + hi * 8 + e) * 4;
                                    s0[e] = lane_bcast(svl, src);
                                    s1[e] = lane_bcast(svl, src + 16);
                                }
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi) {
                                    f32x4 va, vb;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        va[e] = acc[ni][mi][c * 8 + e] * s0[e] * inv[mi];
                                        vb[e] = acc[ni][mi][c * 8 + 4 + e] * s1[e] * inv[mi];
                                    }
                                    const f16x4 ha = __builtin_convertvector(va, f16x4), hb = __builtin_convertvector(vb, f16x4);
                                    const f16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), f16x4);
                                    const f16x4 lb = __builtin_convertvector(vb - __builtin_convertvector(hb, f32x4), f16x4);
                                    unsigned char *dst = blkp[mi] + (ni * 2 + c) * 1024;  // k-step 2 ni + c; plane 1 at + 4096
                                    if (live[mi]) {
                                        *reinterpret_cast<u32x4 *>(dst) =
                                            __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
                                        *reinterpret_cast<u32x4 *>(dst + 4096) =
                                            __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
                                    }
                                }
                            }
                    }
                }
            } else {
                // SWAP: accumulator block [ni][mi] holds out[m][n] with lane = weight row position r32 of block ni (logical
                // column pl_perm(r32)) and registers = 16 token rows in the MFMA C order, register r -> row (r & 3) +
                // 8 (r >> 2) + 4 hi: registers 8 j .. 8 j + 7 are exactly one 16-byte chunk (16-key step j, half hi) of the
                // attention kernel's packed V operand (attn_h2.hip: [plane][j][d half][h][d 32][8 halfs]).
                static_assert(EPI == PL_VPACK || !SWAP, "the swapped orientation exists for the packed-V epilogue");
                if (n_w0 < p.N) {
                    const int nblk = p.v_ntok >> 5;
                    float vb = p.v_bound;
                    if (p.v_bound_dev) vb *= p.v_bound_dev[(p.v_rpc > 0 ? (bm * TM) / p.v_rpc : 0) * p.v_bound_stride];
                    const float sv = h2_scale_of_bound(vb);
                    const float *abase = ascl ? ascl : p.w_scale;
                    float has = ascl ? 1.0f : 0.0f;
                    int64_t mmax = ascl ? p.M - 1 : 0;
                    asm volatile("" : "+v"(abase), "+v"(has), "+v"(mmax));
                    const float *mbase = p.fold_stats ? p.fold_stats : p.w_scale;   // row means (times un = 0 without a fold)
                    int64_t mumax = p.fold_stats ? p.M - 1 : 0;
                    asm volatile("" : "+v"(mbase), "+v"(mumax));
                    // centred operand: plain columns (n >= fold_cols) get + mean u_n, LayerNorm-folded columns (n < fold_cols: the
                    // window attention's V is projected from LN(x), attention.py:262-272) get rstd * (...) + b_n
                    const bool vfold = p.fold_stats && n_w0 < p.fold_cols;  // wave-uniform (fold_cols % 64 == 0)
                    float swn[NI], un[NI], bn[NI];
                    int vcol[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        vcol[ni] = n_w0 + ni * 32 + pl_perm(r32);    // column inside V: head vcol / 64, d half (vcol / 32) & 1
                        swn[ni] = p.w_scale[vcol[ni]] * ascl_c * sv;
                        un[ni] = (p.fold_stats && !vfold && p.fold_u) ? p.fold_u[vcol[ni]] * sv : 0.0f;
                        bn[ni] = (vfold && p.fold_b) ? p.fold_b[vcol[ni]] * sv : 0.0f;
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int64_t row = m_w0 + mi * 32;               // first token of this 32-token block
                        // row factors: lane r32 loads those of row + r32 (coalesced), registers pick theirs by lane shuffles
                        const int64_t rl = row + r32;
                        float sa_l = abase[rl < mmax ? rl : mmax] * has + (1.0f - has);
                        const float mu_l = mbase[2 * (rl < mumax ? rl : mumax)];
                        const float rs_l = mbase[2 * (rl < mumax ? rl : mumax) + (p.fold_stats ? 1 : 0)];
                        if (vfold) sa_l *= rs_l;   // rstd_m (acc sa sw): one more rounding than the fp32 epilogue's order, inside the split's own
                        float sa_r[16], mu_r[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int src = (r & 3) + 8 * (r >> 2) + 4 * hi;  // row of accumulator register r inside the block
                            sa_r[r] = __shfl(sa_l, src);
                            mu_r[r] = __shfl(mu_l, src);
                        }
                        const int64_t rc = row < p.M ? row : p.M - 1;
                        const int64_t seq = rc / p.v_ntok;
                        const int blk = (int)(rc - seq * p.v_ntok) >> 5;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            unsigned char *dst = p.vp + ((seq * p.v_heads + (vcol[ni] >> 6)) * nblk + blk) * 8192 +
                                                 ((((vcol[ni] >> 5) & 1) * 2 + hi) * 32 + (vcol[ni] & 31)) * 16;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                f32x4 pa, pb;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    pa[e] = acc[ni][mi][8 * j + e] * (sa_r[8 * j + e] * swn[ni]) + (mu_r[8 * j + e] * un[ni] + bn[ni]);
                                    pb[e] = acc[ni][mi][8 * j + 4 + e] * (sa_r[8 * j + 4 + e] * swn[ni]) + (mu_r[8 * j + 4 + e] * un[ni] + bn[ni]);
                                }
                                const f16x4 ha = __builtin_convertvector(pa, f16x4), hb = __builtin_convertvector(pb, f16x4);
                                const f16x4 la = __builtin_convertvector(pa - __builtin_convertvector(ha, f32x4), f16x4);
                                const f16x4 lb = __builtin_convertvector(pb - __builtin_convertvector(hb, f32x4), f16x4);
                                if (row < p.M) {
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
        }
        // Every memory operation of the epilogue is complete before the next tile's K loop: a load that the compiler's
        // waitcnt pass still sees "pending" on some path (e.g. one issued inside a wave-uniform branch) would otherwise
        // make it drain vmcnt -- and with it the LDS-DMA ring -- in front of the fragment reads of EVERY K step.  The wait
        // costs the store latency once per tile; the kernel is power-bound, not stall-bound (profiles/r03_gemm_limiter_probe.txt).
        if constexpr ((DBG & 32) == 0) PL_WAIT_VM(0);
        zero_acc();
    }
    if (p.cycles && blockIdx.x == 0 && tid == 0) {
        p.cycles[0] = __builtin_amdgcn_s_memtime() - t_start;
    }
}

}  // namespace omnitok
