// Launchers and operand packers of the plane x plane GEMM (gemm_pl.h).
#include "gemm_pl.h"

namespace omnitok {

extern int g_gemm_gn;
int g_pl_stagger = 0;  // "pl_stagger": start delay step of persistent workgroups (~1 us units), 0 = off
// "temporal_kernel": the fused temporal stage's GEMMs (epilogues 6 / 7) on 1 = gemm_plt_kernel (two workgroups per CU), 0 = gemm_pl_kernel
int g_temporal_kernel = 1;
int g_pl_cfg = 0;  // "pl_cfg": 0 by size (pl_auto_plan) | 1 256x256 | 2 128(n)x256(m) | 5 128x128 | 6 128x64 (see launch_pl)

// ---- weight rows -> scaled fp16 planes, rows permuted inside groups of 32 (pl_perm) ------------------------------
// w' = w * 2^(14 - x) with max|w_n| = m 2^x (m in [0.5, 1)); scale[n] = 2^(x - 14) (by LOGICAL row).  Rows beyond N up
// to n_pad are zero.  One wave per physical row.
__global__ __launch_bounds__(256) void pl_pack_weight_kernel(const float *__restrict__ w, int64_t ldw, int N, int K, int n_pad,
                                                             _Float16 *__restrict__ out, float *__restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int prow = blockIdx.x * 4 + (threadIdx.x >> 6);  // physical row
    if (prow >= n_pad) return;
    const int n = (prow & ~31) + pl_perm(prow & 31);       // logical row stored here
    if (n >= N) {
        for (int k = lane; k < K; k += 64) {
            out[pl_offset(prow, k, 0, K)] = (_Float16)0.0f;
            out[pl_offset(prow, k, 1, K)] = (_Float16)0.0f;
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
        out[pl_offset(prow, k, 0, K)] = h;
        out[pl_offset(prow, k, 1, K)] = l;
    }
}

// ---- activation rows -> planes (not permuted), one power-of-two scale per row or one static scale -----------------
// a' = a * s, s = 2^-e with bound * s in (2^14, 2^15]; a_scale[m] = 1 / s.  Rows [M, m_pad) are zero.
// Workgroup = 64 rows; thread (row = t >> 2, q = t & 3) owns k group q of every 32-k block.
__global__ __launch_bounds__(256) void pl_pack_rows_kernel(const float *__restrict__ x, int64_t ldx, int64_t M, int K,
                                                           unsigned char *__restrict__ planes, float *__restrict__ a_scale,
                                                           float static_bound) {
    const int t = threadIdx.x, row = t >> 2, q = t & 3;
    const int64_t m = (int64_t)blockIdx.x * 64 + row;
    const bool live = m < M;
    const float *xr = x + (live ? m : 0) * ldx;
    float s;
    if (static_bound > 0.0f) {
        s = h2_scale_of_bound(static_bound);
    } else {
        float mx = 0.0f;
        if (live)
            for (int k = q * 8; k < K; k += 32) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(xr + k), v1 = *reinterpret_cast<const f32x4 *>(xr + k + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
            }
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        s = h2_scale_of_bound(mx);
    }
    if (a_scale && q == 0 && live) a_scale[m] = 1.0f / s;
    const int kblocks = K >> 5;
    unsigned char *blk = planes + (int64_t)blockIdx.x * kblocks * 8192 + q * 1024 + row * 16;
    for (int kb = 0; kb < kblocks; ++kb) {
        f32x4 v0 = f32x4{0, 0, 0, 0}, v1 = v0;
        if (live) {
            v0 = *reinterpret_cast<const f32x4 *>(xr + kb * 32 + q * 8);
            v1 = *reinterpret_cast<const f32x4 *>(xr + kb * 32 + q * 8 + 4);
        }
        v0 *= s;
        v1 *= s;
        const f16x4 h0 = __builtin_convertvector(v0, f16x4), h1 = __builtin_convertvector(v1, f16x4);
        const f16x4 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x4), f16x4);
        const f16x4 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x4), f16x4);
        *reinterpret_cast<u32x4 *>(blk + (int64_t)kb * 8192) =
            __builtin_bit_cast(u32x4, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
        *reinterpret_cast<u32x4 *>(blk + (int64_t)kb * 8192 + 4096) =
            __builtin_bit_cast(u32x4, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}

template <int EPI, bool SWAP, typename C>
static int launch_pl_cfg(PlParams p, hipStream_t stream) {
    int n_cu = 0;
    if (int rc = current_device_cus(&n_cu)) return rc;
    constexpr int LDS = C::LDS + (EPI == PL_ROWLN ? 2 * C::WN * C::TM * 4 : 0) +
                        ((EPI == PL_ROWLN || EPI == PL_F32 || EPI == PL_UNPATCH) && C::EPI_T && !(EPI == PL_ROWLN && C::SCR_IN_RING)
                             ? C::NW * 4096 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_pl_kernel<EPI, SWAP, C>), LDS)) return rc;
    const int64_t r_end = p.row_end > 0 ? p.row_end : p.M;
    OT_CHECK_ARG(p.row_begin % C::TM == 0 && p.row_begin < r_end && r_end <= p.M, "gemm_pl: row range [%lld, %lld) of %lld rows on %d-row tiles",
                 (long long)p.row_begin, (long long)r_end, (long long)p.M, C::TM);
    const int64_t nbm = (r_end - p.row_begin + C::TM - 1) / C::TM;
    const int nbn = (p.N + C::TN - 1) / C::TN;
    const int64_t nt = nbm * nbn;
    OT_CHECK_ARG(nt < (1ll << 31), "gemm_pl: grid too large");
    p.bm0 = (int)(p.row_begin / C::TM);
    p.nbm = (int)nbm;
    p.nbn = nbn;
    p.ntiles = (int)nt;
    p.gn = g_gemm_gn > 0 ? g_gemm_gn : 8;
    p.stagger = g_pl_stagger;
    int wg_per_cu = (160 * 1024) / LDS;
    const int by_waves = (C::NI * C::MI > 8) ? 1 : (C::NI * C::MI <= 2 ? 16 : 8) / C::NW;   // (only decides persistent vs one tile per workgroup)
    if (wg_per_cu > by_waves) wg_per_cu = by_waves;
    if (wg_per_cu < 1) wg_per_cu = 1;
    // two small workgroups per CU: one tile each (the hardware dispatcher overlaps one's epilogue with the other's K loop);
    // one big workgroup per CU: persistent
    const int64_t cap = (wg_per_cu > 1 || C::LOOP >= 1) ? nt : (int64_t)n_cu;   // (the pipelined loop is a one-tile-per-workgroup loop)
    const int grid = (int)(nt < cap ? nt : cap);
    hipLaunchKernelGGL((gemm_pl_kernel<EPI, SWAP, C>), dim3(grid), dim3(C::NT), LDS, stream, p);
    OT_LAUNCH_CHECK("gemm_pl");
    return OMNITOK_OK;
}

int launch_plt_tscore(PlParams p, hipStream_t stream);  // gemm_plt.hip
int launch_plt_tpv(PlParams p, hipStream_t stream);

// Tile configurations ("pl_cfg" / omnitok_pl_gemm.cfg; 0 = by size, pl_auto_cfg below).  Every configuration runs the same
// per-accumulator product order (K steps of 16, three plane products per step), so results do not depend on it bit for bit
// (tests/test_gpu_gemm_pl.py); what changes is how many workgroups a launch has and how much of a CU one of them owns:
//   1  256 x 256 (n x m), 8 waves, persistent, one workgroup per CU    -- calls that fill the chip with such tiles
//   2  128 x 256, 4 waves, two per CU                                  -- (A/B arm)
//   5  128 x 128, 4 waves of 64 x 64, two per CU                       -- r06: mid-size calls (4x the tiles of 1)
//   6  128 x  64, 4 waves of 64 x 32, up to four per CU                -- r06: small calls (16x the tiles of 1)
// PL_ROWLN owns whole rows (TN = 512, 8 waves x 64 columns in every configuration: the row statistics are summed in one
// order): 1 -> 128 rows, 6 -> 64 rows, 7 -> 32 rows.  PL_UNPATCH needs the 8-wave row-major epilogue: 1 -> 256 x 256, 6 -> 256 x 64.
using PlBig = PlCfg<4, 2, 4>;
// Thin tiles behind the same 2-step ring as the big ones: 64 / 48 KiB of LDS, so two / three workgroups share a CU and fill each
// other's barrier and fragment-read bubbles.  Deeper rings (DMA 4 / 6 steps ahead, 96 KiB, one workgroup per CU) measured SLOWER
// at every size from 5120 rows up and equal at 1024 (profiles/r06_pl_small_tiles.txt): a thin wave tile has 2 MFMAs per
// group to cover an LDS round trip with, co-resident workgroups are what hides it, not prefetch distance.
using PlMid = PlCfg<2, 2, 4, 2, 0, 2, 2>;      // 128 x 128
using PlSmall = PlCfg<2, 2, 4, 2, 0, 2, 1>;    // 128 x  64
// Closed by measurement this round (profiles/r06_pl_small_tiles.txt; the arms stay in measurement builds, -DOMNITOK_PL_MEASUREMENT_BUILDS):
// a lone 128 x 64 workgroup runs its K loop at ~900 cycles per 16-k step for 192 cycles of MFMA issue (ablation: MFMAs + fragment reads
// alone 570, DMA issue 220, waits + barrier the rest).  FIVE restructurings, all bit-identical to the shipped tiles, all within 5 % of
// the same time at 1024 rows:  deeper rings (DMA 4 / 6 steps ahead: arms 9 / 8);  a loop that reads all of step s + 1's fragments under
// step s's MFMAs and spreads the DMA of step s + R over their gaps (PlCfg LOOP_ = 1: 35 / 36);  one barrier per TWO steps (LOOP_ = 2:
// 37 / 38);  operands through registers instead of LDS-DMA, 4 / 6 steps in flight per wave (LOOP_ = 3: 39 / 40 / 41);  tiles small
// enough that every CU gets one (128 x 32, 64 x 32: 10 / 11).  So it is neither LDS latency, nor barrier count, nor memory-level
// parallelism, nor fetch rate per CU alone; what does help a thin tile is a second / third workgroup on its CU.
#ifdef OMNITOK_PL_MEASUREMENT_BUILDS
using PlMidD = PlCfg<2, 2, 6, 4, 0, 2, 2, 1>;
using PlSmallD = PlCfg<2, 2, 8, 6, 0, 2, 1, 1>;
using PlMid1 = PlCfg<2, 2, 4, 2, 0, 2, 2, 1>;
using PlSmall1 = PlCfg<2, 2, 4, 2, 0, 2, 1, 1>;
using PlSmall3 = PlCfg<2, 2, 2, 4, 0, 2, 1, 3>;   // 128 x 64, operands through registers, 4 steps in flight (arm 39); 40: 6 steps; 41: 128 x 128
using PlSmall3b = PlCfg<2, 2, 2, 6, 0, 2, 1, 3>;
using PlMid3 = PlCfg<2, 2, 2, 3, 0, 2, 2, 3>;
using PlSmall2 = PlCfg<2, 2, 8, 2, 0, 2, 1, 2>;   // 128 x 64, ring of 8, one barrier per two K steps (arm 37); 38: ring of 6
using PlSmall2b = PlCfg<2, 2, 6, 2, 0, 2, 1, 2>;
using PlTiny = PlCfg<2, 1, 4, 2, 0, 2, 1>;     // 128 x 32, two waves
using PlTiny64 = PlCfg<1, 1, 4, 2, 0, 2, 1>;   //  64 x 32, one wave
#endif

// Which configuration(s) a launch takes (results do not depend on it).  From the sweeps in profiles/r06_pl_small_tiles.txt:
//  * 256 x 256 tiles win once they fill >= 3/4 of the CUs; below that 128 x 128 while there are >= n_cu of those, else 128 x 64
//    (PL_ROWLN: the tallest of 128 / 64 / 32 rows that still gives every CU a tile; its full-row workgroups stream the whole weight,
//    so the tall tile stays ahead down to half a round -- 20 480 rows: 64 us on 160 tiles of 128 rows against 72 us on 320 of 64);
//  * TAIL SCHEDULE: when the big tiles make r >= 1 full rounds and a last round that leaves >= 5 % of the launch's CU-rounds idle
//    (N = 512 at 40 960 rows: 320 tiles = 1.25 rounds; C5's 69 632 rows: 2.125), the full rounds run on big tiles and the rows of
//    the last round on thin ones in a second launch behind it ("pl_tail" 1, default).  Per-tile K order is the same in every
//    configuration, so the split is invisible in the bits.
struct PlPlan {
    int cfg;             // configuration of rows [0, split) (all rows when split == 0)
    int64_t split;       // 0: one launch
    int tail_cfg;
};
int g_pl_tail = 1;

static int pl_thin_cfg(int epi, int64_t M, int N, int n_cu) {
    auto tiles = [&](int tm, int tn) { return ((M + tm - 1) / tm) * (int64_t)((N + tn - 1) / tn); };
    if (epi == PL_ROWLN) return (tiles(64, 512) > n_cu || tiles(32, 512) > n_cu) ? 6 : 7;
    if (epi == PL_UNPATCH) return 6;
    return tiles(128, 128) >= n_cu ? 5 : 6;
}

static PlPlan pl_auto_plan(int epi, int64_t M, int N, int64_t row_align) {
    int n_cu = 0;
    if (current_device_cus(&n_cu) != OMNITOK_OK || n_cu <= 0) n_cu = 256;
    const int tm = epi == PL_ROWLN ? 128 : 256, tn = epi == PL_ROWLN ? 512 : 256;
    const int64_t nbm = (M + tm - 1) / tm;
    const int nbn = (N + tn - 1) / tn;
    const int64_t t = nbm * nbn;
    const bool enough = epi == PL_ROWLN ? t * 2 >= n_cu : t * 4 >= 3 * (int64_t)n_cu;
    if (!enough) return PlPlan{pl_thin_cfg(epi, M, N, n_cu), 0, 0};
    const int64_t r = t / n_cu, rem = t % n_cu;
    if (g_pl_tail && r >= 1 && rem > 0 && (n_cu - rem) * 20 >= (int64_t)n_cu * (r + 1)) {
        // rows of the full rounds: whole row tiles, and whole units of `row_align` rows (sequences / clips / videos an epilogue
        // addresses by division: a launch boundary inside one is fine for the arithmetic but keeps the checks simple)
        int64_t rows_main = (r * n_cu / nbn) * tm;
        if (row_align > 1) rows_main -= rows_main % row_align;
        if (rows_main >= tm && rows_main % tm == 0 && rows_main < M)
            return PlPlan{1, rows_main, pl_thin_cfg(epi, M - rows_main, N, n_cu)};
    }
    return PlPlan{1, 0, 0};
}

template <int EPI>
static int launch_pl(const PlParams &p, int cfg, hipStream_t stream) {
    if constexpr (EPI == PL_VPACK) {
        switch (cfg) {
            case 5: return launch_pl_cfg<EPI, true, PlMid>(p, stream);
            case 6: return launch_pl_cfg<EPI, true, PlSmall>(p, stream);
            default: return launch_pl_cfg<EPI, true, PlBig>(p, stream);
        }
    } else if constexpr (EPI == PL_QKPACK) {
        switch (cfg) {
            case 5: return launch_pl_cfg<EPI, false, PlMid>(p, stream);
            case 6: return launch_pl_cfg<EPI, false, PlSmall>(p, stream);
            default: return launch_pl_cfg<EPI, false, PlBig>(p, stream);
        }
    } else if constexpr (EPI == PL_UNPATCH) {
        switch (cfg) {
            case 6: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 3, 2, 0, 2, 1>>(p, stream);
            default: return launch_pl_cfg<EPI, false, PlBig>(p, stream);
        }
    } else if constexpr (EPI == PL_TSCORE || EPI == PL_TPV) {
        // 320 x 128 tiles, 4 waves of (2 x 5 blocks): one wave per SIMD, 160 accumulator registers, 7 DMA pieces per wave and step
        // (the first form, 320 x 256 tiles with 4 x 5 blocks per wave, needed 320 accumulator registers and 9 pieces: 2.6x slower,
        // profiles/r05_temporal_fused.txt); ring of 4 x 28 KiB
        // gemm_plt_kernel addresses its output planes and the softmax weights with 32-bit offsets: calls beyond 4 GiB of either
        // (more than ~400 clips of 17 x 256 x 256 in one call) take the first form below
        const bool fits32 = (int64_t)p.M * (EPI == PL_TPV ? p.cp_kblocks * 32 : 0) * 4 < (1ll << 32) && (int64_t)p.t_nseq * p.t_heads * 160 < (1ll << 32);
        if (g_temporal_kernel >= 1 && fits32) return EPI == PL_TSCORE ? launch_plt_tscore(p, stream) : launch_plt_tpv(p, stream);
        return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 0, 2, 5>>(p, stream);
    } else if constexpr (EPI == PL_ROWLN) {
        switch (cfg) {  // 512 (= N) columns: a workgroup owns whole rows
            case 6: return launch_pl_cfg<EPI, false, PlCfg<8, 1, 4, 3, 0, 2, 2>>(p, stream);  // 64 rows
            case 7: return launch_pl_cfg<EPI, false, PlCfg<8, 1, 4, 3, 0, 2, 1>>(p, stream);  // 32 rows
            default: return launch_pl_cfg<EPI, false, PlCfg<8, 1, 3>>(p, stream);             // 128 rows
        }
    } else {
        switch (cfg) {
            case 2: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 3>>(p, stream);
            case 5: return launch_pl_cfg<EPI, false, PlMid>(p, stream);
            case 6: return launch_pl_cfg<EPI, false, PlSmall>(p, stream);
#ifdef OMNITOK_PL_MEASUREMENT_BUILDS
            case 10: return launch_pl_cfg<EPI, false, PlTiny>(p, stream);
            case 11: return launch_pl_cfg<EPI, false, PlTiny64>(p, stream);
            case 8: return launch_pl_cfg<EPI, false, PlSmallD>(p, stream);
            case 9: return launch_pl_cfg<EPI, false, PlMidD>(p, stream);
            case 35: return launch_pl_cfg<EPI, false, PlMid1>(p, stream);
            case 36: return launch_pl_cfg<EPI, false, PlSmall1>(p, stream);
            case 39: return launch_pl_cfg<EPI, false, PlSmall3>(p, stream);
            case 40: return launch_pl_cfg<EPI, false, PlSmall3b>(p, stream);
            case 41: return launch_pl_cfg<EPI, false, PlMid3>(p, stream);
            case 37: return launch_pl_cfg<EPI, false, PlSmall2>(p, stream);
            case 38: return launch_pl_cfg<EPI, false, PlSmall2b>(p, stream);
            case 21: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 1, 2, 1>>(p, stream);   // 128 x 64: no vmcnt wait
            case 22: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 2, 2, 1>>(p, stream);   // no barrier
            case 23: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 3, 2, 1>>(p, stream);   // neither
            case 24: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 4, 2, 1>>(p, stream);   // no DMA in the K loop
            case 27: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 7, 2, 1>>(p, stream);   // MFMAs + fragment reads only
            case 28: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 8, 2, 1>>(p, stream);   // no epilogue  // wrong-result ablation builds of profiles/r03_gemm_limiter_probe.txt
            case 3: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 3>>(p, stream);
            case 4: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 0, 4, 4>>(p, stream);
            case 31: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 3, 1>>(p, stream);
            case 13: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 3, 3>>(p, stream);
            case 14: return launch_pl_cfg<EPI, false, PlCfg<2, 2, 4, 2, 4, 4, 4>>(p, stream);
            case 33: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 3, 5>>(p, stream);
            case 17: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 3, 7>>(p, stream);
            case 18: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 2, 8>>(p, stream);
            case 19: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 2, 16>>(p, stream);
            case 20: return launch_pl_cfg<EPI, false, PlCfg<4, 2, 4, 2, 32>>(p, stream);
#endif
            default: return launch_pl_cfg<EPI, false, PlBig>(p, stream);
        }
    }
}

// forced_cfg > 0: that configuration for every row; 0: the plan of pl_auto_plan (one launch, or full rounds + tail)
template <int EPI>
static int pl_run(const PlParams &p, int forced_cfg, hipStream_t stream) {
    if (forced_cfg > 0) return launch_pl<EPI>(p, forced_cfg, stream);
    const PlPlan plan = pl_auto_plan(EPI, p.M, p.N, 1);
    if (plan.split == 0) return launch_pl<EPI>(p, plan.cfg, stream);
    PlParams a = p, b = p;
    a.row_begin = 0;
    a.row_end = plan.split;
    b.row_begin = plan.split;
    b.row_end = p.M;
    if (int rc = launch_pl<EPI>(a, plan.cfg, stream)) return rc;
    return launch_pl<EPI>(b, plan.tail_cfg, stream);
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int64_t omnitok_pl_planes_bytes(int64_t rows, int K, int row_pad) {
    if (rows < 0 || K <= 0 || K % 32 || row_pad <= 0 || row_pad % 64) return -1;
    const int64_t rp = (rows + row_pad - 1) / row_pad * row_pad;
    return rp * K * 4;
}

extern "C" float omnitok_pl_unscale(float bound) { return 1.0f / h2_scale_of_bound(bound); }

extern "C" int omnitok_pl_pack_weight(const float *w, int64_t ldw, int N, int K, int n_pad, void *planes, float *scale,
                                      omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(w && planes && scale && N > 0 && K > 0 && K % 32 == 0 && n_pad >= N && n_pad % 64 == 0 && N % 32 == 0,
                 "pl_pack_weight: bad arguments (K %% 32, N %% 32, n_pad %% 64)");
    hipLaunchKernelGGL(pl_pack_weight_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, w, ldw, N, K, n_pad,
                       static_cast<_Float16 *>(planes), scale);
    OT_LAUNCH_CHECK("pl_pack_weight");
    return OMNITOK_OK;
}

extern "C" int omnitok_pl_pack_rows(const float *x, int64_t ldx, int64_t M, int K, int64_t m_pad, void *planes, float *a_scale,
                                    float static_bound, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(x && planes && M >= 0 && K > 0 && K % 32 == 0 && m_pad >= M && m_pad % 64 == 0 && ldx % 4 == 0 && aligned16(x) &&
                     (a_scale || static_bound > 0.0f),
                 "pl_pack_rows: bad arguments");
    if (m_pad == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(pl_pack_rows_kernel, dim3((unsigned)(m_pad / 64)), dim3(256), 0, stream, x, ldx, M, K,
                       static_cast<unsigned char *>(planes), a_scale, static_bound);
    OT_LAUNCH_CHECK("pl_pack_rows");
    return OMNITOK_OK;
}

extern "C" int omnitok_gemm_pl(const omnitok_pl_gemm *g, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(g && g->a && g->w && g->w_scale, "gemm_pl: null pointer");
    OT_CHECK_ARG(g->M >= 0 && g->N > 0 && g->K >= 32 && g->K % 32 == 0 && g->N % 32 == 0, "gemm_pl: bad sizes M=%lld N=%d K=%d",
                 (long long)g->M, g->N, g->K);
    if (g->M == 0) return OMNITOK_OK;
    PlParams p{};
    p.a = static_cast<const unsigned char *>(g->a);
    p.a2 = static_cast<const unsigned char *>(g->a2);
    p.a_scale = g->a_scale;
    p.a2_scale = g->a2_scale;
    p.a_scale_const = g->a_scale_const > 0.0f ? g->a_scale_const : 1.0f;
    p.a2_scale_const = g->a2_scale_const > 0.0f ? g->a2_scale_const : 1.0f;
    p.a_split_n = g->a2 ? g->a_split_n : 0;
    p.w = static_cast<const unsigned char *>(g->w);
    p.w_scale = g->w_scale;
    p.bias = g->bias;
    p.residual = g->residual;
    p.ldr = g->ldr;
    p.c = g->c;
    p.ldc = g->ldc;
    p.c2 = g->c2;
    p.ldc2 = g->ldc2;
    p.c_split_n = g->c2 ? g->c_split_n : 0;
    p.cp = static_cast<unsigned char *>(g->out_planes);
    p.cp_kblocks = g->out_planes_k / 32;
    p.cp_scale = g->out_bound > 0.0f ? h2_scale_of_bound(g->out_bound) : 1.0f;
    p.M = g->M;
    p.N = g->N;
    p.K = g->K;
    OT_CHECK_ARG(g->k_valid >= 0 && g->k_valid <= g->K, "gemm_pl: k_valid=%d outside [0, K]", g->k_valid);
    p.nk = g->k_valid > 0 ? (g->k_valid + 15) / 16 : g->K / 16;
    p.cycles = g->debug_cycles;
    if (g->fold_stats) {
        OT_CHECK_ARG(g->fold_cols >= 0 && g->fold_cols % 64 == 0 && (g->fold_cols >= g->N || g->fold_u) &&
                         (g->epilogue == PL_F32 || g->epilogue == PL_QKPACK || g->epilogue == PL_VPACK || g->epilogue == PL_TSCORE ||
                          g->epilogue == PL_TPV),
                     "gemm_pl: a centred operand needs fold_cols %% 64 == 0, the row sums fold_u for the columns >= fold_cols, "
                     "and epilogue 0, 3, 4, 6 or 7");
        p.fold_stats = g->fold_stats;
        p.fold_b = g->fold_b;
        p.fold_u = g->fold_u;
        p.fold_cols = g->fold_cols;
    }
    if (g->a_rpg > 0) {
        OT_CHECK_ARG(g->a_rpg % 256 == 0 && g->a_goff % 256 == 0 && g->a_gstride % 64 == 0 && g->M % g->a_rpg == 0 && !g->a2 &&
                         !g->a_scale && !g->fold_stats,
                     "gemm_pl: operand row map needs a_rpg, a_goff %% 256 == 0, a_gstride %% 64 == 0, whole groups, and one static "
                     "operand scale (per-row scales / statistics are indexed by the unmapped row)");
        p.a_rpg = g->a_rpg;
        p.a_gstride = g->a_gstride;
        p.a_goff = g->a_goff;
    }
    const int cfg = g->cfg > 0 ? g->cfg : (g_pl_cfg > 0 ? g_pl_cfg : 0);   // 0: by size (pl_auto_plan)
    OT_CHECK_ARG(p.a_split_n % 256 == 0 && p.c_split_n % 256 == 0, "gemm_pl: split columns must be multiples of 256");
    switch (g->epilogue) {
        case PL_F32:
            OT_CHECK_ARG(g->c && g->ldc % 4 == 0 && aligned16(g->c) && (!g->residual || (g->ldr % 4 == 0 && aligned16(g->residual))),
                         "gemm_pl: fp32 output must be 16-byte aligned with ldc %% 4 == 0");
            return pl_run<PL_F32>(p, cfg, stream);
        case PL_GEGLU:
            OT_CHECK_ARG(g->out_planes && g->N % 64 == 0 && g->out_planes_k == g->N / 2 && g->out_bound > 0.0f,
                         "gemm_pl: GEGLU needs out_planes with out_planes_k == N / 2 and a bound of the hidden");
            return pl_run<PL_GEGLU>(p, cfg, stream);
        case PL_VPACK:
            OT_CHECK_ARG(g->vp && aligned16(g->vp) && g->heads > 0 && g->N == g->heads * 64 && g->n_tokens > 0 &&
                             g->n_tokens % 32 == 0 && g->M % g->n_tokens == 0 && g->v_bound > 0.0f && !g->a2 &&
                             (!g->v_bound_dev || (g->rows_per_clip > 0 && g->rows_per_clip % 256 == 0)),
                         "gemm_pl: packed V needs vp, N == heads * 64, whole sequences of n_tokens %% 32 == 0 rows, a bound and "
                         "rows_per_clip %% 256 == 0");
            p.vp = static_cast<unsigned char *>(g->vp);
            p.v_ntok = g->n_tokens;
            p.v_heads = g->heads;
            p.v_bound = g->v_bound;
            p.v_bound_dev = g->v_bound_dev;
            p.v_bound_stride = g->v_bound_stride > 0 ? g->v_bound_stride : 1;
            p.v_rpc = g->v_bound_dev ? g->rows_per_clip : 0;
            return pl_run<PL_VPACK>(p, cfg, stream);
        case PL_QKPACK:
            OT_CHECK_ARG(g->qp && g->kp && aligned16(g->qp) && aligned16(g->kp) && g->heads > 0 && g->qk_k0 == g->heads * 64 &&
                             g->N == 2 * g->qk_k0 && g->n_tokens > 0 && g->n_tokens % 32 == 0 && g->M % g->n_tokens == 0 &&
                             (g->q_scale == nullptr) == (g->k_scale == nullptr) && g->q_bound > 0.0f && g->k_bound > 0.0f &&
                             (g->rope_cos == nullptr) == (g->rope_sin == nullptr),
                         "gemm_pl: packed Q | K needs qp, kp, N == 2 * heads * 64, whole sequences of n_tokens %% 32 == 0 rows, "
                         "both q / k scale vectors (or neither: no l2 normalisation) and bounds");
            p.qp = static_cast<unsigned char *>(g->qp);
            p.kp = static_cast<unsigned char *>(g->kp);
            p.qk_k0 = g->qk_k0;
            p.qk_ntok = g->n_tokens;
            p.qk_heads = g->heads;
            p.cosT = g->rope_cos;
            p.sinT = g->rope_sin;
            p.q_scale = g->q_scale;
            p.k_scale = g->k_scale;
            p.q_mul = g->q_mul;
            p.sq = h2_scale_of_bound(g->q_bound);
            p.sk = h2_scale_of_bound(g->k_bound);
            return pl_run<PL_QKPACK>(p, cfg, stream);
        case PL_UNPATCH:
            OT_CHECK_ARG(g->c && aligned16(g->c) && g->up_p == 8 && g->up_pt > 0 && g->up_t > 0 && g->up_C > 0 &&
                             g->up_H % g->up_p == 0 && g->up_W % (32 * g->up_p) == 0 &&
                             g->N == g->up_C * g->up_pt * g->up_p * g->up_p && g->up_f0 >= 0 &&
                             g->up_f0 + g->up_t * g->up_pt <= g->up_F &&
                             g->M % ((int64_t)g->up_t * (g->up_H / g->up_p) * (g->up_W / g->up_p)) == 0 && !g->residual && !g->c2,
                         "gemm_pl: the un-patchify epilogue needs p == 8, W %% 256 == 0, N == C pt p p and whole videos of t x (H / p) x (W / p) patches");
            p.up_C = g->up_C; p.up_F = g->up_F; p.up_H = g->up_H; p.up_W = g->up_W; p.up_f0 = g->up_f0; p.up_t = g->up_t;
            p.up_pt = g->up_pt; p.up_p = g->up_p;
            return pl_run<PL_UNPATCH>(p, cfg, stream);
        case PL_ROWLN:
            OT_CHECK_ARG(g->N == 512 && g->c && g->ldc % 4 == 0 && aligned16(g->c) && g->out_planes && g->out_planes_k == g->N &&
                             g->out_bound > 0.0f && g->ln_gamma && (!g->residual || (g->ldr % 4 == 0 && aligned16(g->residual))) &&
                             !g->c2 && !g->a2,
                         "gemm_pl: the LayerNorm epilogue needs N == 512 (full-row tiles), c, out_planes (K = N), gamma and a bound");
            p.ln_gamma = g->ln_gamma;
            p.ln_beta = g->ln_beta;
            p.ln_eps = g->ln_eps;
            return pl_run<PL_ROWLN>(p, cfg, stream);
        case PL_TSCORE:
        case PL_TPV: {
            // rows: [tile of 64 sequences][32-sequence half][5 time steps][32 sequences] (omnitok_stats_pack_temporal)
            OT_CHECK_ARG(g->M % 320 == 0 && g->t_nseq > 0 && g->M == (int64_t)(g->t_nseq + 63) / 64 * 320 && g->t_heads > 0 &&
                             g->tp && aligned16(g->tp) && g->fold_stats && g->a_scale && !g->a2 && g->a_rpg == 0,
                         "gemm_pl: the temporal epilogues need M = ceil(nseq / 64) * 320 permuted rows, row statistics, per-row "
                         "scales and the P buffer");
            p.tp = g->tp;
            p.t_nseq = g->t_nseq;
            p.t_heads = g->t_heads;
            p.t_alibi = g->t_alibi;
            p.t_scale = g->q_mul > 0.0f ? g->q_mul : 1.0f;
            if (g->epilogue == PL_TSCORE) {
                OT_CHECK_ARG(g->N == g->t_heads * 128 && g->q_scale && g->k_scale && g->fold_u,
                             "gemm_pl: temporal scores need N = heads * 128 ([q_h | k_h] columns), q / k scale vectors and fold_u");
                p.q_scale = g->q_scale;
                p.k_scale = g->k_scale;
                return launch_pl<PL_TSCORE>(p, cfg, stream);
            }
            OT_CHECK_ARG(g->N == g->t_heads * 64 && g->t_heads % 2 == 0 && g->out_planes && aligned16(g->out_planes) &&
                             g->out_planes_k == g->N && g->t_out_scale && g->v_bound > 0.0f &&
                             (!g->v_bound_dev || g->t_seqs_per_clip > 0),
                         "gemm_pl: temporal P.V needs N = heads * 64, out_planes (K = N), t_out_scale and a bound of |v|");
            p.t_out_scale = g->t_out_scale;
            p.v_bound = g->v_bound;
            p.v_bound_dev = g->v_bound_dev;
            p.v_bound_stride = g->v_bound_stride > 0 ? g->v_bound_stride : 1;
            p.t_seqs_per_clip = g->t_seqs_per_clip > 0 ? g->t_seqs_per_clip : g->t_nseq;
            return launch_pl<PL_TPV>(p, cfg, stream);
        }
        default:
            set_error("gemm_pl: unsupported epilogue %d", g->epilogue);
            return OMNITOK_ERR_INVALID;
    }
}
