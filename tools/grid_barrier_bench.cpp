// Cost of a software grid barrier on MI355X (one persistent workgroup per CU): the number the persistent LM decode kernel
// of DESIGN.md section (i) stands or falls with (120 barriers per token).  Three arms, N barriers each:
//   bare      release fence + atomic add (agent scope) by one lane per workgroup, bounded spin on an acquire load
//   stream    the same with 64 KiB of coalesced weight reads per workgroup between two barriers (a decode phase streams
//             ~8-16 MiB over 256 CUs)
//   prefetch  like stream, but the NEXT phase's reads are issued BEFORE waiting at the barrier (they do not depend on it)
// Every spin is bounded (SPIN_LIMIT polls, then an error flag and exit): a barrier that never completes must not hang
// the box.   build: tools/build_grid_barrier_bench.sh     run: tools/_bin/grid_barrier_bench [--n 2000] [--wg 256]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));             \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int SPIN_LIMIT = 1 << 20;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);  // agent scope by default for global atomics
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_LIMIT) {
                atomicExch(err, 1);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

// two-level form: workgroups of one XCD (blockIdx % 8, each XCD has its own L2) meet on their own counter line, the last
// arrival of each XCD adds to the global counter, everybody polls the global one: 32 + 8 serialized atomics per barrier
// instead of 256 on one address
__device__ __forceinline__ bool grid_barrier_2l(unsigned *counters, unsigned it1, unsigned nwg, int *err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7u, per = (nwg + 7u - xcd) / 8u;  // workgroups on this XCD
        const unsigned prev = __atomic_fetch_add(counters + 32 * (1 + xcd), 1u, __ATOMIC_ACQ_REL);
        if (prev + 1 == it1 * per) __atomic_fetch_add(counters, 1u, __ATOMIC_RELEASE);
        const unsigned nx = nwg < 8u ? nwg : 8u;
        int spins = 0;
        while (__hip_atomic_load(counters, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < it1 * nx) {
            if (++spins > SPIN_LIMIT) {
                atomicExch(err, 1);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

template <int ARM>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned *counter, int *err, int n, const f32x4 *w, int64_t w_v4,
                                                      float *sink) {
    const unsigned nwg = gridDim.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // 64 KiB per workgroup and phase = 4096 float4 = 16 per thread
    auto issue = [&](int phase, f32x4 (&v)[16]) {
        const int64_t base = ((int64_t)phase * nwg + blockIdx.x) * 4096 % (w_v4 - 4096);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = w[base + i * 256 + threadIdx.x];
    };
    f32x4 nxt[16];
    if (ARM == 2) issue(0, nxt);
    for (int it = 0; it < n; ++it) {
        if (ARM == 1) {
            f32x4 v[16];
            issue(it, v);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i];
        } else if (ARM == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += nxt[i];
            issue(it + 1, nxt);  // next phase's weights are on their way while this workgroup waits
        }
        if (ARM == 3) {
            if (!grid_barrier_2l(counter, (unsigned)(it + 1), nwg, err)) break;
        } else if (!grid_barrier(counter, (unsigned)(it + 1) * nwg, err))
            break;
        if (*(volatile int *)err) break;
    }
    if (ARM == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += nxt[i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x] = acc[0];
}

int main(int argc, char **argv) {
    int n = 2000, wg = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--n")) n = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--wg")) wg = atoi(argv[++i]);
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    if (wg <= 0) wg = prop.multiProcessorCount;  // one workgroup per CU: all co-resident, the barrier cannot deadlock
    if (wg > prop.multiProcessorCount) {
        fprintf(stderr, "more workgroups than CUs: not guaranteed co-resident, refusing\n");
        return 1;
    }
    unsigned *counter;
    int *err;
    float *sink;
    f32x4 *w;
    const int64_t w_bytes = 512ll << 20, w_v4 = w_bytes / 16;  // larger than L2 + MALL: the reads stream from HBM
    CK(hipMalloc(&counter, 4 * 32 * 9));  // global counter + one 128-byte line per XCD
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&sink, wg * 4));
    CK(hipMalloc(&w, w_bytes));
    CK(hipMemset(w, 0, w_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[4] = {"bare", "stream 64 KiB/wg between barriers", "stream, next phase requested before the wait",
                            "bare, two-level (per-XCD counter, then global)"};
    for (int arm = 0; arm < 4; ++arm) {
        float best = 1e30f;
        int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counter, 0, 4 * 32 * 9));
            CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(e0));
            if (arm == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(wg), dim3(256), 0, 0, counter, err, n, w, w_v4, sink);
            if (arm == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(wg), dim3(256), 0, 0, counter, err, n, w, w_v4, sink);
            if (arm == 2) hipLaunchKernelGGL(barrier_kernel<2>, dim3(wg), dim3(256), 0, 0, counter, err, n, w, w_v4, sink);
            if (arm == 3) hipLaunchKernelGGL(barrier_kernel<3>, dim3(wg), dim3(256), 0, 0, counter, err, n, w, w_v4, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (herr) break;
            if (ms < best) best = ms;
        }
        if (herr) {
            printf("%-48s spin limit hit (error flag set)\n", names[arm]);
            continue;
        }
        const double us = best * 1e3 / n;
        const double gbs = (arm == 1 || arm == 2) ? (double)wg * 65536.0 / (us * 1e-6) / 1e9 : 0.0;
        printf("%-48s %7.2f us per phase+barrier (%d workgroups, %d barriers)%s", names[arm], us, wg, n, (arm == 1 || arm == 2) ? "" : "\n");
        if (arm == 1 || arm == 2) printf("   %.0f GB/s streamed\n", gbs);
    }
    return 0;
}
