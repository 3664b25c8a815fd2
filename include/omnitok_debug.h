/*
 * omnitok_debug.h -- measurement-only entry points of libomnitok.so (not part of the drop-in boundary of omnitok.h):
 * kernel timeline stamps and the bare-MFMA peak probe behind profiles/.  Nothing in the product path calls them.
 */
#ifndef OMNITOK_DEBUG_H
#define OMNITOK_DEBUG_H

#include "omnitok.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement only: device buffer of 4 x 96 int64; workgroup 0 of the persistent GEMM stores
 * s_memtime stamps (3 per K-step per wave: stream start, stream end, loop end). NULL = off. */
int omnitok_debug_set_gemm_trace(long long *dev_ptr);
/* Measurement only: pure v_mfma_f32_32x32x2_f32 stream (4 accumulators per wave, operands from
 * `in`[4096]) to find the sustained fp32-MFMA ceiling of the chip. out[blocks*256]. */
int omnitok_debug_mfma_peak(const float *in, float *out, int blocks, int iters, int lds_bytes,
                            long long *clk, omnitok_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNITOK_DEBUG_H */
