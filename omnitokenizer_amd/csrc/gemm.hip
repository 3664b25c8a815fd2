// fp32 GEMM  c = a . w^T (+ epilogue) on v_mfma_f32_32x32x2_f32 -- every nn.Linear on the
// OmniTokenizer encode/decode path (reference attention.py:164,167,271,287,386-393;
// omnitokenizer.py:810,819,1007,1013).  87 % of the path's FLOPs run through this kernel.
//
// Design (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators = 64 accumulator VGPRs), BK = 32 per step.
//  * global -> registers -> LDS staging (T14 split: the next K-step's loads are issued before the
//    MFMAs of the current one, written to the other LDS buffer after them; one barrier per step);
//  * LDS rows padded to 36 floats so that the four ds_read_b128 of a fragment are conflict-free
//    (16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots: (9*row) mod 16);
//  * K is consumed in a lane-permuted order: MFMA step s of a K-step multiplies k = s (lanes
//    0-31) and k = 16+s (lanes 32-63) for both operands, so every lane reads 16 *contiguous*
//    floats of its row (4 x ds_read_b128) instead of 16 scalar LDS reads;
//  * workgroup ids are remapped (bijectively) so that each XCD sweeps whole rows of output tiles:
//    the 128xK A panel is then served by that XCD's own L2;
//  * epilogues (bias, residual add, GEGLU, leaky-relu) are fused on the accumulator registers.
// fp32 MFMA is bitwise a k-ordered fmaf chain, so the result differs from the reference's MKL
// GEMM only by summation order (~1e-7 relative).
#include "gemm_common.h"

namespace omnitok {

// tuning knobs (omnitok_set_option, A/B measurements): "gemm_variant" 0 = one 128x128 tile per
// workgroup, 1 / 2 = persistent workgroups with the fully interleaved MFMA stream (128x128 x 4 waves /
// 256x128 x 8 waves), +4 = force; "gemm_lds_pad_kb" extra dynamic LDS per workgroup (variants 0, 1).
int g_gemm_variant = 1;
int g_gemm_lds_pad_kb = 0;
int g_gemm_gn = 8;     // "gemm_gn": tile columns per column group
int g_gemm_small = 1;  // "gemm_small": 64x128 tiles when the 128x128 tiling cannot fill the chip
long long *g_gemm_trace = nullptr;

template <int FLAGS, bool NEDGE>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;

    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int64_t bm;
    int bn;
    tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);

    // ---- loader mapping: thread -> 4 rows x one float4 column of the 128x32 tile ----------
    const int lrow = tid >> 3, lc4 = tid & 7;
    const float *ap[4];
    const float *wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t gr = bm * BM + lrow + 32 * i;
        if (gr > p.M - 1) gr = p.M - 1;
        int64_t ar = gr;
        if (p.a_rpg > 0) ar = (gr / p.a_rpg) * p.a_stride + p.a_off + (gr % p.a_rpg);
        ap[i] = p.a + ar * p.lda + lc4 * 4;
        int wr = bn * BN + lrow + 32 * i;
        if (wr > p.N - 1) wr = p.N - 1;
        wp[i] = p.w + (int64_t)wr * p.ldw + lc4 * 4;
    }
    const int st_off = lrow * LDT + lc4 * 4;  // + i*32*LDT

    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const f32x4 *>(ap[i] + k0);
            rb[i] = *reinterpret_cast<const f32x4 *>(wp[i] + k0);
        }
    };
    auto lstore = [&](int buf) {
        float *As = smem + buf * 2 * TILE_FLOATS;
        float *Bs = As + TILE_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(As + st_off + i * 32 * LDT) = ra[i];
            *reinterpret_cast<f32x4 *>(Bs + st_off + i * 32 * LDT) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int a_frag_off = (wm * 64 + r32) * LDT + hi * 16;
    const int b_frag_off = (wn * 64 + r32) * LDT + hi * 16;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const float *As = smem + buf * 2 * TILE_FLOATS;
        const float *Bs = As + TILE_FLOATS;
        f32x4 af[2][4], bf[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                af[mb][j] = *reinterpret_cast<const f32x4 *>(As + a_frag_off + mb * 32 * LDT + 4 * j);
                bf[mb][j] = *reinterpret_cast<const f32x4 *>(Bs + b_frag_off + mb * 32 * LDT + 4 * j);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mb][j][e], bf[nb][j][e],
                                                                          acc[mb][nb], 0, 0, 0);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    gemm_epilogue<FLAGS, NEDGE>(p, acc, bm, bn, wm, wn, r32, hi);
}


// Small-problem variant (launch-latency regime, e.g. one 256x256 image = 1024 tokens): when the
// 128x128 tiling gives fewer workgroups than CUs, the time of a GEMM is the serial MFMA chain of one
// wave.  64 x (64*NB) tiles with 32 x (32*NB) wave tiles shorten that chain 2x (NB = 2) or 4x (NB = 1) and
// multiply the workgroup count; with NB = 2 the GEGLU value/gate pairing (two adjacent 32-column
// blocks) still lives inside one wave.
constexpr int SM_BM = 64;

template <int FLAGS, int NB>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_small(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SBN = 64 * NB;                  // tile columns
    constexpr int STAGE = (SM_BM + SBN) * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nbn = (p.N + SBN - 1) / SBN;
    const int64_t bm = blockIdx.x / nbn;
    const int bn = blockIdx.x % nbn;

    const int lrow = tid >> 3, lc4 = tid & 7;
    const float *ap[2];
    const float *wp[2 * NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t gr = bm * SM_BM + lrow + 32 * i;
        if (gr > p.M - 1) gr = p.M - 1;
        int64_t ar = gr;
        if (p.a_rpg > 0) ar = (gr / p.a_rpg) * p.a_stride + p.a_off + (gr % p.a_rpg);
        ap[i] = p.a + ar * p.lda + lc4 * 4;
    }
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) {
        int wr = bn * SBN + lrow + 32 * i;
        if (wr > p.N - 1) wr = p.N - 1;
        wp[i] = p.w + (int64_t)wr * p.ldw + lc4 * 4;
    }
    const int st_off = lrow * LDT + lc4 * 4;
    f32x4 ra[2], rb[2 * NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ap[i] + k0);
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(wp[i] + k0);
    };
    auto lstore = [&](int buf) {
        float *As = smem + buf * STAGE;
        float *Bs = As + SM_BM * LDT;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(As + st_off + i * 32 * LDT) = ra[i];
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) *reinterpret_cast<f32x4 *>(Bs + st_off + i * 32 * LDT) = rb[i];
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    const int nk = p.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int a_frag_off = (wm * 32 + r32) * LDT + hi * 16;
    const int b_frag_off = (wn * 32 * NB + r32) * LDT + hi * 16;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const float *As = smem + buf * STAGE;
        const float *Bs = As + SM_BM * LDT;
        f32x4 af[4], bf[NB][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            af[j] = *reinterpret_cast<const f32x4 *>(As + a_frag_off + 4 * j);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[nb][j] = *reinterpret_cast<const f32x4 *>(Bs + b_frag_off + nb * 32 * LDT + 4 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], bf[nb][j][e], acc[nb], 0, 0, 0);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    const int64_t row0 = bm * SM_BM + wm * 32;
    if constexpr ((FLAGS & OMNITOK_GEMM_GEGLU) != 0) {
        static_assert(NB == 2 || !(FLAGS & OMNITOK_GEMM_GEGLU), "GEGLU pairs two 32-column blocks per wave");
        const int ocol = (bn * 2 + wn) * 32 + r32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + mfma32_row(r, hi);
            if (row < p.M) p.c[row * p.ldc + ocol] = gelu_erf(acc[NB - 1][r]) * acc[0][r];
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = bn * SBN + wn * 32 * NB + nb * 32 + r32;
            const bool colok = col < p.N;
            float bv = 0.0f;
            if constexpr (FLAGS & OMNITOK_GEMM_BIAS) bv = colok ? p.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + mfma32_row(r, hi);
                if (row < p.M && colok) {
                    float v = acc[nb][r];
                    if constexpr (FLAGS & OMNITOK_GEMM_BIAS) v += bv;
                    if constexpr (FLAGS & OMNITOK_GEMM_LEAKY) v = v > 0.0f ? v : 0.1f * v;
                    if constexpr (FLAGS & OMNITOK_GEMM_RESIDUAL) v += p.residual[row * p.ldr + col];
                    p.c[row * p.ldc + col] = v;
                }
            }
        }
    }
}

template <int FLAGS, int NB>
static int launch_small(const GemmParams &p, hipStream_t stream) {
    constexpr int SBN = 64 * NB;
    constexpr int lds = 2 * (SM_BM + SBN) * LDT * 4;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_f32_mfma_small<FLAGS, NB>), lds)) return rc;
    const int64_t nwg = ((p.M + SM_BM - 1) / SM_BM) * ((p.N + SBN - 1) / SBN);
    hipLaunchKernelGGL((gemm_f32_mfma_small<FLAGS, NB>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
    OT_LAUNCH_CHECK("gemm_f32_mfma_small");
    return OMNITOK_OK;
}

// Persistent variant: workgroups walk the tile list (tile = blockIdx.x + i * gridDim.x, then
// XCD-remapped); the global->register->LDS pipeline runs ACROSS tile boundaries, so per tile there
// is no launch, no address prologue and no exposed first-load latency -- only the epilogue's issue
// time sits between the last MFMA of one tile and the first of the next.
// WM = wave rows: WM = 2 -> 128x128 tile, 4 waves, two workgroups per CU;
//                 WM = 4 -> 256x128 tile, 8 waves (two per SIMD, same barrier domain), one per CU.
template <int FLAGS, int WM, bool TRACE>
__global__ __launch_bounds__(128 * WM, 2) void gemm_f32_mfma_persistent(GemmParams p) {
    constexpr int NT = 128 * WM;          // threads
    constexpr int TM = 64 * WM;           // tile rows
    constexpr int RPP = NT / 8;           // tile rows covered by one loader pass
    constexpr int NA = TM / RPP;          // loader passes over A (= 4)
    constexpr int NB = BN / RPP;          // loader passes over B (4 or 2)
    constexpr int NP = NA + NB;           // staged float4 pieces per thread per K-step
    constexpr int A_FLOATS = TM * LDT, STAGE_FLOATS = (TM + BN) * LDT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nk = p.K / BK;
    if ((int)blockIdx.x >= p.ntiles) return;

    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    const int lrow = tid >> 3, lc4 = tid & 7;
    const float *ap[NA];
    const float *wp[NB];
    auto set_ptrs = [&](int i) {
        const int lid = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, p.ntiles);
        int64_t tbm;
        int tbn;
        tile_coords(lid, p.nbm, p.nbn, p.gn, tbm, tbn);
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            int64_t gr = tbm * TM + lrow + RPP * r;
            if (gr > p.M - 1) gr = p.M - 1;
            int64_t ar = gr;
            if (p.a_rpg > 0) ar = (gr / p.a_rpg) * p.a_stride + p.a_off + (gr % p.a_rpg);
            ap[r] = p.a + ar * p.lda + lc4 * 4;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            int wr = tbn * BN + lrow + RPP * r;
            if (wr > p.N - 1) wr = p.N - 1;
            wp[r] = p.w + (int64_t)wr * p.ldw + lc4 * 4;
        }
    };
    const int st_off = lrow * LDT + lc4 * 4;
    f32x4 ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ap[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(wp[i] + k0);
    };
    auto lstore = [&](int buf) {
        float *As = smem + buf * STAGE_FLOATS;
        float *Bs = As + A_FLOATS;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4 *>(As + st_off + i * RPP * LDT) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4 *>(Bs + st_off + i * RPP * LDT) = rb[i];
    };
    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    zero_acc();
    const int a_frag_off = (wm * 64 + r32) * LDT + hi * 16;
    const int b_frag_off = (wn * 64 + r32) * LDT + hi * 16;

    // Measured with s_memtime stamps (tools/gemm_trace.py): while one wave streams MFMAs its SIMD
    // partner's non-MFMA instructions are not issued, so two 4-wave workgroups on a CU end up taking
    // turns K-step by K-step and everything a wave does OUTSIDE its own MFMA stream is exposed time.
    // So every LDS / global instruction is embedded in the stream, between groups of 4 MFMAs:
    //   steps 0-7  (fragments F0 of K-step g): ds_read F1(g), ds_write of K-step g+1 (fetched one
    //              K-step ago; all stores before the first new load), global loads of K-step g+2
    //   lgkmcnt(0) + s_barrier            (orders LDS only; K-step g+1 is now complete in LDS)
    //   steps 8-15 (fragments F1): ds_read F0(g+1)
    // Hazards: tile g+1 is stored before barrier(g) and first read after it; buffer g&1 is
    // overwritten (tile g+2) after barrier(g+1) >= every read of tile g (issued before barrier(g)).
    int tn = 0;
    auto stamp = [&]() {
        if constexpr (TRACE)
            if (p.trace && blockIdx.x == 0 && lane == 0 && tn < 96) p.trace[wave * 96 + tn++] = __builtin_readcyclecounter();
    };
    f32x4 fa0[2][2], fb0[2][2], fa1[2][2], fb1[2][2];  // [row block][j] fragments of the two halves
    auto frag_piece = [&](f32x4 (&fa)[2][2], f32x4 (&fb)[2][2], int buf, int half, int piece) {
        const float *As = smem + buf * STAGE_FLOATS;
        const float *Bs = As + A_FLOATS;
        const int mb = (piece >> 1) & 1, j = piece & 1;
        if (piece < 4)
            fa[mb][j] = *reinterpret_cast<const f32x4 *>(As + a_frag_off + mb * 32 * LDT + 8 * half + 4 * j);
        else
            fb[mb][j] = *reinterpret_cast<const f32x4 *>(Bs + b_frag_off + mb * 32 * LDT + 8 * half + 4 * j);
    };
    // store/load are compile-time after unrolling: a RUNTIME condition around a load makes hipcc
    // branch around it and wait vmcnt(0) per element (cdna guide 5, trap (c)); past the end of the
    // tile list the stream simply re-fetches valid data and stores it to the idle buffer
    auto stage_piece = [&](int buf, int piece, bool store, bool load, int k0) {
        float *As = smem + buf * STAGE_FLOATS;
        float *Bs = As + A_FLOATS;
        if (piece < NA) {
            if (store) *reinterpret_cast<f32x4 *>(As + st_off + piece * RPP * LDT) = ra[piece];
            if (load) ra[piece] = *reinterpret_cast<const f32x4 *>(ap[piece] + k0);
        } else if (piece < NP) {
            const int i = piece - NA;
            if (store) *reinterpret_cast<f32x4 *>(Bs + st_off + i * RPP * LDT) = rb[i];
            if (load) rb[i] = *reinterpret_cast<const f32x4 *>(wp[i] + k0);
        }
    };
    // one K-step of the stream: fragments of LDS buffer `buf`, stores the staged K-step into the
    // other buffer and fetches the K-step at offset k0 of the current load pointers
    auto kstep = [&](int buf, int k0) {
        stamp();
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int j = step >> 2, e = step & 3;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[mb][j][e], fb0[nb][j][e], acc[mb][nb], 0,
                                                                      0, 0);
            if (step < 4) {  // F1 reads early in the half: long landed when the second half starts
                frag_piece(fa1, fb1, buf, 1, 2 * step);
                frag_piece(fa1, fb1, buf, 1, 2 * step + 1);
            }
            // all LDS stores (steps 0-3) before the first new global load (steps 4-7), so that the
            // vmcnt wait in front of a store never covers a load issued in this K-step
            if (step < 4) {
                stage_piece(buf ^ 1, 2 * step, true, false, k0);
                stage_piece(buf ^ 1, 2 * step + 1, true, false, k0);
            } else {
                stage_piece(buf ^ 1, 2 * (step - 4), false, true, k0);
                stage_piece(buf ^ 1, 2 * (step - 4) + 1, false, true, k0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int j = step >> 2, e = step & 3;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[mb][j][e], fb1[nb][j][e], acc[mb][nb], 0,
                                                                      0, 0);
            if (step < 4) {
                frag_piece(fa0, fb0, buf ^ 1, 0, 2 * step);
                frag_piece(fa0, fb0, buf ^ 1, 0, 2 * step + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();
    };

    // prologue: K-steps 0 and 1 of the first tile (nk is even and >= 2: the launcher checks)
    set_ptrs(0);
    gload(0);
    lstore(0);
    gload(BK);  // stays in registers until the middle of the first K-step
    lds_barrier();
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) frag_piece(fa0, fb0, 0, 0, pc);

    for (int ti = 0; ti < my_tiles; ++ti) {
        // K-steps 0 .. nk-3 fetch K-steps 2 .. nk-1 of the same tile
        for (int k = 0; k < nk - 2; ++k) kstep(k & 1, (k + 2) * BK);
        // the last two fetch K-steps 0 and 1 of the next tile (of this tile again when it is the last)
        if (ti + 1 < my_tiles) set_ptrs(ti + 1);
        kstep(0, 0);   // k = nk-2 (even): LDS buffer 0
        kstep(1, BK);  // k = nk-1
        const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
        int64_t ebm;
        int ebn;
        tile_coords(lid, p.nbm, p.nbn, p.gn, ebm, ebn);
        // the shared epilogue addresses 64x64 wave tiles in 128-row units
        gemm_epilogue<FLAGS, false>(p, acc, ebm * (WM / 2) + (wm >> 1), ebn, wm & 1, wn, r32, hi);
        zero_acc();
    }
}

template <int FLAGS, int WM>
static int launch_persistent(GemmParams p, int n_cu, hipStream_t stream) {
    constexpr int TM = 64 * WM;
    const int lds = 2 * (TM + BN) * LDT * 4 + (WM == 2 ? g_gemm_lds_pad_kb * 1024 : 0);
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_f32_mfma_persistent<FLAGS, WM, false>), lds))
        return rc;
    if constexpr (FLAGS == 0)
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_f32_mfma_persistent<0, WM, true>), lds))
            return rc;
    p.nbm = (int)((p.M + TM - 1) / TM);
    const int64_t ntiles = (int64_t)p.nbm * p.nbn;
    OT_CHECK_ARG(ntiles < (1ll << 31), "gemm grid too large");
    p.ntiles = (int)ntiles;
    const int wg_per_cu = WM == 2 ? 2 : 1;
    if (FLAGS == 0 && p.trace) {  // instrumented build of the plain variant (tools/gemm_trace.py)
        hipLaunchKernelGGL((gemm_f32_mfma_persistent<0, WM, true>), dim3(wg_per_cu * n_cu), dim3(128 * WM), lds,
                           stream, p);
    } else {
        hipLaunchKernelGGL((gemm_f32_mfma_persistent<FLAGS, WM, false>), dim3(wg_per_cu * n_cu), dim3(128 * WM), lds,
                           stream, p);
    }
    OT_LAUNCH_CHECK("gemm_f32_mfma_persistent");
    return OMNITOK_OK;
}

template <int FLAGS>
static int launch_gemm(GemmParams p, hipStream_t stream) {
    int n_cu = 0;
    if (int rc = current_device_cus(&n_cu)) return rc;
    const int64_t nbm = (p.M + BM - 1) / BM;
    const int64_t nwg = nbm * p.nbn;
    OT_CHECK_ARG(nwg < (1ll << 31), "gemm grid too large");
    p.ntiles = (int)nwg;
    p.nbm = (int)nbm;
    p.trace = g_gemm_trace;
    // variant: 0 one 128x128 tile per workgroup; 1 persistent 128x128 (two 4-wave workgroups per CU);
    // 2 persistent 256x128 (one 8-wave workgroup per CU).  Persistent kernels only when the problem
    // fills the chip more than once; bit 2 (+4) forces the variant whatever the size (tests).
    int variant = g_gemm_variant & 3;
    const bool force = (g_gemm_variant & 4) != 0;
    if (variant && nwg <= 2 * n_cu && !force) variant = 0;
    const int nk = p.K / BK;
    if (nk < 2 || (nk & 1)) variant = 0;  // the persistent stream is unrolled over K-step pairs
    if (p.N % 64) variant = 0;            // persistent kernels use the unguarded buffer epilogue
    // measured at C3: the 8-wave 256x128 workgroup is ahead for the wide GEGLU GEMM (131 vs 128 TF),
    // the two 4-wave 128x128 workgroups for the N <= 1536 projections (134-140 vs 131-140 TF)
    if (variant == 1 && !force && (FLAGS & OMNITOK_GEMM_GEGLU)) variant = 2;
    if (variant == 2) return launch_persistent<FLAGS, 4>(p, n_cu, stream);
    if (variant == 1) return launch_persistent<FLAGS, 2>(p, n_cu, stream);
    if (nwg < n_cu && g_gemm_small && !force) {
        // fewer 128x128 tiles than CUs: 64x128 tiles (32x64 per wave, half the serial MFMA chain); if even
        // those leave most of the chip idle, 64x64 tiles (32x32 per wave, a quarter of the chain)
        if constexpr (!(FLAGS & OMNITOK_GEMM_GEGLU)) {
            const int64_t nwg128 = ((p.M + SM_BM - 1) / SM_BM) * p.nbn;
            if (nwg128 < n_cu / 2 || g_gemm_small == 2) return launch_small<FLAGS, 1>(p, stream);
        }
        return launch_small<FLAGS, 2>(p, stream);
    }
    const int lds = GEMM_LDS_BYTES + g_gemm_lds_pad_kb * 1024;
    const bool nedge = (p.N % 64) != 0;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_f32_mfma<FLAGS, false>), lds)) return rc;
    if constexpr (!(FLAGS & OMNITOK_GEMM_GEGLU))
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_f32_mfma<FLAGS, true>), lds)) return rc;
    if constexpr (!(FLAGS & OMNITOK_GEMM_GEGLU)) {
        if (nedge) {
            hipLaunchKernelGGL((gemm_f32_mfma<FLAGS, true>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
            OT_LAUNCH_CHECK("gemm_f32_mfma");
            return OMNITOK_OK;
        }
    }
    hipLaunchKernelGGL((gemm_f32_mfma<FLAGS, false>), dim3((unsigned)nwg), dim3(256), lds, stream, p);
    OT_LAUNCH_CHECK("gemm_f32_mfma");
    return OMNITOK_OK;
}

__global__ void pack_geglu_kernel(const float *w1, int inner, int K, int inner_pad, float *out) {
    // out row pr: block = pr / 64, within = pr % 64; within < 32 -> value row j = block*32+within,
    // else gate row j = block*32 + within-32 (source row inner + j)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)2 * inner_pad * K;
    if (idx >= total) return;
    const int pr = (int)(idx / K), k = (int)(idx % K);
    const int blk = pr >> 6, within = pr & 63;
    const int j = blk * 32 + (within & 31);
    float v = 0.0f;
    if (j < inner) v = w1[(int64_t)((within < 32) ? j : inner + j) * K + k];
    out[idx] = v;
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_gemm(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                            const float *residual, int64_t ldr, float *c, int64_t ldc, int64_t M, int N,
                            int K, int flags, int64_t a_rows_per_group, int64_t a_group_stride,
                            int64_t a_group_offset, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(a && w && c, "gemm: null pointer");
    OT_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return OMNITOK_OK;
    OT_CHECK_ARG(K % BK == 0, "gemm: K=%d must be a multiple of %d (pad the weight)", K, BK);
    OT_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && aligned16(a) && aligned16(w),
                 "gemm: operands must be 16-byte aligned with ld %% 4 == 0");
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_BIAS) || bias, "gemm: BIAS flag without bias");
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_RESIDUAL) || residual, "gemm: RESIDUAL flag without residual");
    GemmParams p;
    p.a = a; p.w = w; p.bias = bias; p.residual = residual; p.c = c;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.nbn = (N + BN - 1) / BN;
    p.a_rpg = a_rows_per_group; p.a_stride = a_group_stride; p.a_off = a_group_offset;
    p.gn = g_gemm_gn > 0 ? g_gemm_gn : 8;
    switch (flags) {
        case 0: return launch_gemm<0>(p, stream);
        case OMNITOK_GEMM_BIAS: return launch_gemm<OMNITOK_GEMM_BIAS>(p, stream);
        case OMNITOK_GEMM_RESIDUAL: return launch_gemm<OMNITOK_GEMM_RESIDUAL>(p, stream);
        case OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL:
            return launch_gemm<OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL>(p, stream);
        case OMNITOK_GEMM_BIAS | OMNITOK_GEMM_LEAKY:
            return launch_gemm<OMNITOK_GEMM_BIAS | OMNITOK_GEMM_LEAKY>(p, stream);
        case OMNITOK_GEMM_GEGLU:
            OT_CHECK_ARG(N % BN == 0, "gemm: GEGLU packed width %d must be a multiple of %d", N, BN);
            return launch_gemm<OMNITOK_GEMM_GEGLU>(p, stream);
        default:
            set_error("gemm: unsupported epilogue flags %d", flags);
            return OMNITOK_ERR_INVALID;
    }
}

extern "C" int omnitok_pack_geglu_weight(const float *w1, int inner, int K, int inner_pad, float *out,
                                         omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(inner_pad % 64 == 0 && inner_pad >= inner, "pack_geglu: inner_pad %d invalid", inner_pad);
    const int64_t total = (int64_t)2 * inner_pad * K;
    hipLaunchKernelGGL(pack_geglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w1,
                       inner, K, inner_pad, out);
    OT_LAUNCH_CHECK("pack_geglu");
    return OMNITOK_OK;
}
