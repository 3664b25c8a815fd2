// Stand-alone probe (r06): how fast does ONE wave issue v_mfma_f32_32x32x16_f16 as a function of the number of independent
// accumulators it alternates between (the thin GEMM tiles have 2 or 4; the big one 8), with 1 or 2 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain.hip -o tools/_bin/mfma_chain && tools/_bin/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ void chain(float *out, long long *cyc, int iters) {
    __shared__ f16x8 sm[512];
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    f16x8 A, B;
    for (int e = 0; e < 8; ++e) { A[e] = (_Float16)(threadIdx.x * 0.001f + e); B[e] = (_Float16)(e * 0.5f); }
    sm[threadIdx.x] = A;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (LDS) {   // a fragment read per round, as the GEMM's K loop has
            A = sm[(threadIdx.x + i) & 511];
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
static void run(int threads, const char *tag) {
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((chain<NACC, LDS>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((chain<NACC, LDS>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s NACC %d  %d waves/SIMD: %6.1f cycles per MFMA per wave (s_memtime ticks; 3 x NACC MFMAs per round)\n", tag, NACC, threads / 256,
           (double)c / iters / (3 * NACC));
    hipFree(out); hipFree(cyc);
}

// chip-wide sustained rate: `blocks` workgroups of `threads` threads, NACC accumulators, hipEvent time
template <int NACC>
static void chip(int threads, int blocks) {
    float *out; long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, (size_t)blocks * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<NACC, false>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC, false>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 3.0 * NACC * 32768.0;
    printf("chip: %4d workgroups x %d waves, NACC %d: %.2f ms, %.0f TFLOP/s of v_mfma_f32_32x32x16_f16\n", blocks, threads / 64, NACC, ms, flop / ms / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    chip<4>(256, 256);    // one wave per SIMD
    chip<4>(512, 256);    // two
    chip<4>(1024, 256);   // four
    chip<2>(512, 512);
    for (int thr : {256, 512}) {
        run<1, false>(thr, "mfma only");
        run<2, false>(thr, "mfma only");
        run<4, false>(thr, "mfma only");
        run<8, false>(thr, "mfma only");
        run<2, true>(thr, "mfma + one ds_read per round");
        run<4, true>(thr, "mfma + one ds_read per round");
    }
    return 0;
}
