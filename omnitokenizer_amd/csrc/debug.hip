// Measurement-only kernels (not on the product path): practical fp32-MFMA ceiling of this chip under
// sustained load on non-trivial data, to put the GEMM / attention numbers in context.
#include "common.h"
#include "../../include/omnitok_debug.h"

namespace omnitok {

template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        int iters, long long *clk) {
    const int lane = threadIdx.x & 63;
    const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(threadIdx.x * 8 + i) & 4095];
        b[i] = in[(threadIdx.x * 8 + i + 2048) & 4095];
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + i) & 7], acc[i], 0, 0, 0);
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum + (float)lane;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - t0;  // shader-clock ticks (s_memtime)
        clk[1] = wall_clock64() - w0;                // constant 100 MHz ticks
    }
}

}  // namespace omnitok

using namespace omnitok;

// Runs `blocks` workgroups of 256 threads, each wave issuing iters * 8 * 4 MFMAs (32x32x2 f32).
// lds_bytes of dynamic LDS is requested only to steer occupancy (e.g. 100 KiB -> 1 workgroup/CU).
extern "C" int omnitok_debug_mfma_peak(const float *in, float *out, int blocks, int iters, int lds_bytes,
                                       long long *clk, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(in && out && blocks > 0 && iters > 0, "debug_mfma_peak: bad arguments");
    static int attr = 0;
    if (lds_bytes > attr) {
        OT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_peak_kernel<4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr = lds_bytes;
    }
    hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(blocks), dim3(256), lds_bytes, stream, in, out, iters, clk);
    OT_LAUNCH_CHECK("mfma_peak");
    return OMNITOK_OK;
}
