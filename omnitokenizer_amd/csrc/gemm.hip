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
    int64_t a_rpg, a_stride, a_off;
};

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;

    // XCD-aware bijective remap (block b runs on XCD b % 8)
    const int nwg = gridDim.x;
    int lid;
    {
        const int id = blockIdx.x;
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
    }
    const int64_t bm = lid / p.nbn;
    const int bn = lid % p.nbn;

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

    // ---- epilogue ---------------------------------------------------------------------------
    const int64_t row0 = bm * BM + wm * 64;
    if constexpr (FLAGS & OMNITOK_GEMM_GEGLU) {
        const int ocol = (bn * 2 + wn) * 32 + r32;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + mb * 32 + mfma32_row(r, hi);
                if (row < p.M) p.c[row * p.ldc + ocol] = gelu_erf(acc[mb][1][r]) * acc[mb][0][r];
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

template <int FLAGS>
static int launch_gemm(const GemmParams &p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        OT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f32_mfma<FLAGS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
        attr_set = true;
    }
    const int64_t nbm = (p.M + BM - 1) / BM;
    const int64_t nwg = nbm * p.nbn;
    OT_CHECK_ARG(nwg < (1ll << 31), "gemm grid too large");
    hipLaunchKernelGGL(gemm_f32_mfma<FLAGS>, dim3((unsigned)nwg), dim3(256), GEMM_LDS_BYTES, stream, p);
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
