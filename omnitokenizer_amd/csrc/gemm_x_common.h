// Shared by the split-operand GEMMs (gemm_x3.hip: bf16 x 3 planes, gemm_h2.hip: fp16 x 2 planes):
// buffer-descriptor epilogue of one wave tile.
#pragma once
#include "gemm_common.h"

namespace omnitok {

__device__ __forceinline__ auto x3_rsrc(const float *ptr, int bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi32 << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Fused epilogue of one wave tile (MI*32 rows x NI*32 columns at (row0, col0)); N % 32 == 0.
// Same technique as gemm_epilogue: wave-uniform buffer descriptors bound the rows (loads of rows
// >= M return 0, stores are dropped), so there are no per-element guards.
// SC: every accumulator column block ni is first multiplied by cs[ni] (the power-of-two scales of the fp16x2
// GEMM, exact).
template <int FLAGS, int MI, int NI, bool SC = false>
__device__ __forceinline__ void x3_epilogue(const GemmParams &p, f32x16 (&acc)[MI][NI], int64_t row0, int col0,
                                            int r32, int hi, const float *cs = nullptr) {
    int64_t vr64 = p.M - row0;
    const int vr = vr64 > MI * 32 ? MI * 32 : (vr64 < 0 ? 0 : (int)vr64);
    const int ldc = (int)p.ldc;
    if constexpr (FLAGS & OMNITOK_GEMM_GEGLU) {
        static_assert(NI % 2 == 0 || !(FLAGS & OMNITOK_GEMM_GEGLU), "GEGLU pairs two 32-column blocks per wave");
        const int ocol0 = col0 / 2;
        const int ow = NI * 16;
        const auto c_rs = x3_rsrc(p.c + row0 * p.ldc + ocol0, vr > 0 ? ((vr - 1) * ldc + ow) * 4 : 0);
        const int c_voff = (4 * hi * ldc + r32) * 4;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int np = 0; np < NI / 2; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                {
                    float gate = acc[mi][2 * np + 1][r], val = acc[mi][2 * np][r];
                    if constexpr (SC) {
                        gate *= cs[2 * np + 1];
                        val *= cs[2 * np];
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gelu_erf(gate) * val), c_rs,
                                                          c_voff, ((mi * 32 + mfma32_row(r, 0)) * ldc + np * 32) * 4,
                                                          0);
                }
    } else {
        if (col0 >= p.N) return;
        int vc = p.N - col0;  // valid columns of this wave tile (multiple of 32)
        if (vc > NI * 32) vc = NI * 32;
        const int ldr = (int)p.ldr;
        const auto c_rs = x3_rsrc(p.c + row0 * p.ldc + col0, vr > 0 ? ((vr - 1) * ldc + vc) * 4 : 0);
        const int c_voff = (4 * hi * ldc + r32) * 4;
        if constexpr ((FLAGS & OMNITOK_GEMM_RESIDUAL) != 0 && (MI * NI) % 4 == 0) {
            // Residual tiles: the loads of FOUR 32x32 blocks (64 registers) are issued before the first store of the
            // group.  Loads and stores go through buffer descriptors that may alias (x += ... runs in place), so a
            // block-by-block loop exposes one memory round trip per block: measured as a per-tile overhead worth
            // 36 K-steps on the to_out GEMM (K = 512, 199 TF vs 294 TF for the same shape without residual).
            const auto r_rs = x3_rsrc(p.residual + row0 * p.ldr + col0, vr > 0 ? ((vr - 1) * ldr + vc) * 4 : 0);
            const int r_voff = (4 * hi * ldr + r32) * 4;
            constexpr int NB4 = MI * NI / 4;
#pragma unroll
            for (int g = 0; g < NB4; ++g) {
                float res[4][16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int blk = g * 4 + q, ni = blk / MI, mi = blk % MI;
                    if (ni * 32 < vc) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            res[q][r] = __builtin_bit_cast(
                                float, __builtin_amdgcn_raw_buffer_load_b32(
                                           r_rs, r_voff, ((mi * 32 + mfma32_row(r, 0)) * ldr + ni * 32) * 4, 0));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int blk = g * 4 + q, ni = blk / MI, mi = blk % MI;
                    if (ni * 32 < vc) {
                        float bv = 0.0f;
                        if constexpr (FLAGS & OMNITOK_GEMM_BIAS) bv = p.bias[col0 + ni * 32 + r32];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[mi][ni][r];
                            if constexpr (SC) v *= cs[ni];
                            if constexpr (FLAGS & OMNITOK_GEMM_BIAS) v += bv;
                            if constexpr (FLAGS & OMNITOK_GEMM_LEAKY) v = v > 0.0f ? v : 0.1f * v;
                            v += res[q][r];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, c_voff,
                                                                  ((mi * 32 + mfma32_row(r, 0)) * ldc + ni * 32) * 4, 0);
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            if (ni * 32 >= vc) break;
            float bv = 0.0f;
            if constexpr (FLAGS & OMNITOK_GEMM_BIAS) bv = p.bias[col0 + ni * 32 + r32];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float res[16];  // 16 residual loads in flight
                if constexpr (FLAGS & OMNITOK_GEMM_RESIDUAL) {
                    const auto r_rs =
                        x3_rsrc(p.residual + row0 * p.ldr + col0, vr > 0 ? ((vr - 1) * ldr + vc) * 4 : 0);
                    const int r_voff = (4 * hi * ldr + r32) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[r] = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(
                                       r_rs, r_voff, ((mi * 32 + mfma32_row(r, 0)) * ldr + ni * 32) * 4, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mi][ni][r];
                    if constexpr (SC) v *= cs[ni];
                    if constexpr (FLAGS & OMNITOK_GEMM_BIAS) v += bv;
                    if constexpr (FLAGS & OMNITOK_GEMM_LEAKY) v = v > 0.0f ? v : 0.1f * v;
                    if constexpr (FLAGS & OMNITOK_GEMM_RESIDUAL) v += res[r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, c_voff,
                                                          ((mi * 32 + mfma32_row(r, 0)) * ldc + ni * 32) * 4, 0);
                }
            }
        }
    }
}

}  // namespace omnitok
