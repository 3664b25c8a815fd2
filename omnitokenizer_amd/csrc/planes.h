// Activation operands of the plane x plane GEMM (gemm_pl.h) written by their producers: layout helpers and the
// epilogue that turns an attention kernel's O^T accumulators into plane chunks.
//
// Plane block layout (64 rows x 32 k, 8 KiB): [plane hi|lo][k group (k / 8) % 4][row % 64][k % 8] fp16;
// blocks ordered [row / 64][k / 32].
#pragma once
#include "h2_common.h"

namespace omnitok {

// byte offset of the 16-byte chunk (row, k group kg of k block kb) of plane 0 (plane 1: + 4096)
__device__ __forceinline__ int64_t pl_chunk_offset(int64_t row, int kb, int kg, int kblocks) {
    return ((row >> 6) * kblocks + kb) * 8192 + kg * 1024 + (row & 63) * 16;
}

// split 8 scaled fp32 values into the hi and lo fp16 chunks and store them
__device__ __forceinline__ void pl_store_chunk(unsigned char *dst, f32x4 va, f32x4 vb) {
    const f16x4 ha = __builtin_convertvector(va, f16x4), hb = __builtin_convertvector(vb, f16x4);
    const f16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), f16x4);
    const f16x4 lb = __builtin_convertvector(vb - __builtin_convertvector(hb, f32x4), f16x4);
    *reinterpret_cast<u32x4 *>(dst) = __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
    *reinterpret_cast<u32x4 *>(dst + 4096) = __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
}

// O^T accumulator block of the attention kernels (32 d x 32 queries, lane = query r32, registers = d in the MFMA C
// layout: register 4 g + e of lane half h is d = 8 g + 4 h + e) -> plane chunks of row `row`, k block `kb`
// (k = head * 64 + d).  A chunk is 8 consecutive d = the two halves of a wave: lanes 0-31 take the chunks g = 0, 2,
// lanes 32-63 the chunks g = 1, 3 after one v_permlane32_swap per register pair.  `mul` = 1 / softmax sum etc. times
// the power-of-two operand scale.
__device__ __forceinline__ void pl_store_ot_block(const f32x16 &ot, float mul, unsigned char *planes, int64_t row, int kb,
                                                  int kblocks, int hi) {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        f32x4 va, vb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned x0 = __builtin_bit_cast(unsigned, ot[(2 * gp) * 4 + e] * mul);      // chunk 2 gp:     lo half d 0..3 | hi half d 4..7
            const unsigned x1 = __builtin_bit_cast(unsigned, ot[(2 * gp + 1) * 4 + e] * mul);  // chunk 2 gp + 1
            // r[0]: lanes 0-31 keep x0, lanes 32-63 receive x1 of lanes 0-31; r[1]: lanes 0-31 receive x0 of lanes 32-63,
            // lanes 32-63 keep x1  ->  lanes 0-31 hold chunk 2 gp (own d 0..3, partner's d 4..7), lanes 32-63 chunk 2 gp + 1
            const auto r = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
            va[e] = __builtin_bit_cast(float, (unsigned)r[0]);
            vb[e] = __builtin_bit_cast(float, (unsigned)r[1]);
        }
        pl_store_chunk(planes + pl_chunk_offset(row, kb, 2 * gp + hi, kblocks), va, vb);
    }
}

}  // namespace omnitok
