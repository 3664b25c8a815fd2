// Shared helpers for the gfx950 kernels of the OmniTokenizer encode/decode path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/omnitok.h"

namespace omnitok {

void set_error(const char *fmt, ...);

#define OT_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            omnitok::set_error(__VA_ARGS__);            \
            return OMNITOK_ERR_INVALID;                 \
        }                                               \
    } while (0)

#define OT_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess) {                                                          \
            omnitok::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),    \
                               __FILE__, __LINE__);                                      \
            return OMNITOK_ERR_HIP;                                                      \
        }                                                                                \
    } while (0)

#define OT_LAUNCH_CHECK(name)                                                            \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            omnitok::set_error("launch of %s failed: %s", name, hipGetErrorString(_e));  \
            return OMNITOK_ERR_HIP;                                                      \
        }                                                                                \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) cached per (kernel, device) under a mutex: the attribute
// is per device, so a process-wide flag would leave a second GPU (or a racing second host thread) without it.
int set_max_dynamic_lds(const void *kernel, int bytes);
// multiprocessor count of the current device (cached per device)
int current_device_cus(int *n_cu);

// fills n 32-bit words with `value` by a kernel.  Used instead of hipMemsetAsync on every path that may be captured into a
// HIP graph: a replayed graph's memset nodes were observed to stop taking effect once an eager call had run between two
// replays (ROCm 7.2, tests/test_gpu_e2e.py::test_hip_graph_capture_and_side_stream), kernel nodes do not.
int device_fill_u32(void *ptr, unsigned value, int64_t n_words, hipStream_t stream);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device helpers -----------------------------------------------------------------------
#ifdef __HIPCC__

// DPP controls (gfx9): quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// all-reduce (sum) over each 16-lane row of the wave: 4 full-rate DPP adds, no LDS
__device__ __forceinline__ float row16_allsum(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror
    v += dpp_f32<0x140>(v);  // row_mirror
    return v;
}

// pre_vq arithmetic shared by pre_vq_kernel (vq.hip) and layernorm_prevq_kernel (norm.hip): stated operation by operation
// (no compiler contraction choices), so that the two kernels produce the same bits whatever surrounds the call
__device__ __forceinline__ float prevq_dot4(const float x0, const float x1, const float x2, const float x3, const float w0,
                                            const float w1, const float w2, const float w3) {
#pragma clang fp contract(off)
    const float a = __builtin_fmaf(x0, w0, x1 * w1);
    const float b = __builtin_fmaf(x2, w2, x3 * w3);
    return a + b;
}
// z[0..8) -> z / max(||z||, 1e-12) (F.normalize, reference omnitokenizer.py:252); sequential sum of squares
__device__ __forceinline__ void prevq_l2norm8(float (&z)[8]) {
#pragma clang fp contract(off)
    float ss = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) ss = __builtin_fmaf(z[c], z[c], ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = z[c] / den;
}

__device__ __forceinline__ float row16_allmax(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}

// full-wave (64 lanes) all-reduce sum
__device__ __forceinline__ float wave_allsum(float v) {
    v = row16_allsum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// value held by the partner lane (lane ^ 32)
__device__ __forceinline__ float swap32(float v) { return __shfl_xor(v, 32); }

// A product that must be ROUNDED before it is added to something (the reference's own sequence of fp32 roundings, or results that
// may not depend on which instantiation the compiler fused an fma in): passing it through an empty asm hides the multiply from the
// fma combiner.  `#pragma clang fp contract(off)` is not enough -- HIP compiles with -ffp-contract=fast-honor-pragmas, and the
// backend's combiner (fusion mode "fast") contracts a * b - c * d whatever the instruction flags say once packed fp32 is off
// (found in r06: lm_select's CFG blend came out as one fma).
__device__ __forceinline__ float no_fuse(float x) {
    asm("" : "+v"(x));
    return x;
}

// all-reduce over the two 32-lane halves of a wave (lane l with lane l ^ 32) through v_permlane32_swap (gfx950):
// no LDS crossbar round trip.  r[0] = value of lanes 0-31 in both halves, r[1] = value of lanes 32-63.
__device__ __forceinline__ float halves_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float halves_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// row index inside a 32x32 MFMA C/D tile for accumulator register r of a lane in half `hi`
// (cdna guide section 3: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31)
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

#endif  // __HIPCC__

}  // namespace omnitok
