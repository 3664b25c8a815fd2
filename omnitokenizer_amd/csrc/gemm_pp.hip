// fp32 GEMM  c = a . w^T  on the fp16 matrix cores with BOTH operands already split into fp16 hi|lo planes in the
// block layout of omnitok_h2_pack_weight ("pp" = packed x packed).  No VALU work and no ds_write in the K loop: the
// operands go global -> LDS by global_load_lds_dwordx4 (the layout is lane-linear), a three-stage ring with counted
// vmcnt waits keeps two K-steps of loads in flight across the per-step barrier, and two independent 128x256
// workgroups per CU cover each other's barriers and epilogues.
//
// This is the measurement vehicle for DESIGN.md (h)-1: the h2 GEMM's ablations say that staging the fp32 A operand
// (global load -> LayerNorm / split VALU -> ds_write) is what separates it from the MFMA-only rate; here that path
// does not exist, at the price of a producer writing the planes.  c[m][n] = acc * a_scale[m] * w_scale[n].
#include "h2_common.h"

namespace omnitok {

typedef __attribute__((address_space(3))) void pp_lds_t;
typedef __attribute__((address_space(1))) const void pp_glob_t;

constexpr int PP_TM = 128, PP_TN = 256, PP_STAGES = 3;
constexpr int PP_A_BYTES = 2 * 2 * 2 * 1024;   // 2 row blocks x 2 planes x 2 k-groups x 1 KiB
constexpr int PP_W_BYTES = 4 * 2 * 2 * 1024;
constexpr int PP_STAGE = PP_A_BYTES + PP_W_BYTES;  // 24 KiB
constexpr int PP_LDS = PP_STAGES * PP_STAGE;       // 72 KiB: two workgroups per CU

struct PPParams {
    const unsigned char *a, *w;   // packed planes ([rows/64][K/32] blocks of 8 KiB)
    const float *a_scale, *w_scale;
    float *c;
    int64_t ldc, M;
    int N, K, nbn, ntiles;
};

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt untouched): gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define PP_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))

__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(PPParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pp_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    const int bn = tile % p.nbn;
    const int64_t bm = tile / p.nbn;
    const int kblocks = p.K >> 5, nsteps = p.K >> 4;

    // per step: 24 chunks of 1 KiB (A 8, W 16); wave v moves chunks v, v + 4, ... (6 each)
    const unsigned char *gsrc[6];
    int ldst[6], kgsel[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = wave + 4 * i;                 // 0..23
        const bool is_w = c >= 8;
        const int cc = is_w ? c - 8 : c;            // [blk][plane][kgl]
        const int blk = cc >> 2, pl = (cc >> 1) & 1, kgl = cc & 1;
        const int64_t rb = is_w ? (int64_t)bn * 4 + blk : bm * 2 + blk;
        gsrc[i] = (is_w ? p.w : p.a) + rb * kblocks * 8192 + pl * 4096 + kgl * 1024 + lane * 16;
        ldst[i] = (is_w ? PP_A_BYTES : 0) + cc * 1024;
        kgsel[i] = kgl;
    }
    auto dma = [&](int step, int stage) {  // K-step `step` (16 k): block step >> 1, k-groups 2 (step & 1) + {0, 1}
        const int64_t goff = (int64_t)(step >> 1) * 8192 + (step & 1) * 2048;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((pp_glob_t *)(gsrc[i] + goff), (pp_lds_t *)(pp_smem + stage * PP_STAGE + ldst[i]),
                                             16, 0, 0);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // fragment addresses inside a stage
    const int a_frag = ((wm * 2) * 2 + hi) * 1024 + r32 * 16;                   // + pl * 2048 + mi * 512
    const int w_frag = PP_A_BYTES + ((wn * 2) * 2 * 2 + hi) * 1024 + r32 * 16;  // + (ni >> 1) * 4096 + pl * 2048 + (ni & 1) * 512

    dma(0, 0);
    if (nsteps > 1) dma(1, 1);
    if (nsteps > 1) { PP_WAIT_VM(6); } else { PP_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < nsteps; ++s) {
        const int stage = s % PP_STAGES;
        if (s + 2 < nsteps) dma(s + 2, (s + 2) % PP_STAGES);
        const unsigned char *S = pp_smem + stage * PP_STAGE;
        u32x4 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            ah[mi] = *reinterpret_cast<const u32x4 *>(S + a_frag + mi * 512);
            al[mi] = *reinterpret_cast<const u32x4 *>(S + a_frag + 2048 + mi * 512);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            bh[ni] = *reinterpret_cast<const u32x4 *>(S + w_frag + (ni >> 1) * 4096 + (ni & 1) * 512);
            bl[ni] = *reinterpret_cast<const u32x4 *>(S + w_frag + (ni >> 1) * 4096 + 2048 + (ni & 1) * 512);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[mi]),
                                                                   __builtin_bit_cast(f16x8, bh[ni]), acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mi]),
                                                                   __builtin_bit_cast(f16x8, bl[ni]), acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mi]),
                                                                   __builtin_bit_cast(f16x8, bh[ni]), acc[mi][ni], 0, 0, 0);
            }
        // step s + 1 must have landed before anyone reads it; step s + 2 (just issued) may stay in flight
        if (s + 2 < nsteps) { PP_WAIT_VM(6); } else { PP_WAIT_VM(0); }
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: undo the row scales of both operands
    const int64_t row0 = bm * PP_TM + wm * 64;
    const int col0 = bn * PP_TN + wn * 128;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int col = col0 + ni * 32 + r32;
        const float sw = p.w_scale[col];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + mi * 32 + mfma32_row(r, hi);
                p.c[row * p.ldc + col] = acc[mi][ni][r] * (p.a_scale[row] * sw);
            }
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_gemm_pp(const void *a_planes, const float *a_scale, const void *w_planes, const float *w_scale,
                               float *c, int64_t ldc, int64_t M, int N, int K, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(a_planes && a_scale && w_planes && w_scale && c, "gemm_pp: null pointer");
    OT_CHECK_ARG(M > 0 && M % PP_TM == 0 && N % PP_TN == 0 && K % 32 == 0 && K >= 32,
                 "gemm_pp: needs M %% 128 == 0, N %% 256 == 0, K %% 32 == 0 (measurement kernel)");
    PPParams p;
    p.a = static_cast<const unsigned char *>(a_planes); p.w = static_cast<const unsigned char *>(w_planes);
    p.a_scale = a_scale; p.w_scale = w_scale; p.c = c; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.nbn = N / PP_TN;
    const int64_t nt = (M / PP_TM) * p.nbn;
    OT_CHECK_ARG(nt < (1ll << 31), "gemm_pp: grid too large");
    p.ntiles = (int)nt;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(gemm_pp_kernel), PP_LDS)) return rc;
    hipLaunchKernelGGL(gemm_pp_kernel, dim3((unsigned)nt), dim3(256), PP_LDS, stream, p);
    OT_LAUNCH_CHECK("gemm_pp");
    return OMNITOK_OK;
}
