// GPU box probe: does an LDS-DMA load (global_load_lds_dwordx4, destination = workgroup LDS base + M0) land where a ds_read of the
// same workgroup-relative offset looks, when TWO workgroups share a CU and the second one's LDS block crosses 128 KiB?
// Each workgroup fills its LDS with a marker, DMAs 1 KiB of its own source line to offset `off`, idles (so that neighbours are
// co-resident), then checks the whole block.  Prints, per (LDS bytes per workgroup, offset): workgroups with a wrong / missing
// piece, workgroups with foreign bytes elsewhere in their block, and the raw LDS_ALLOC register of a failing workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_t;
typedef __attribute__((address_space(1))) const void glob_t;

__global__ __launch_bounds__(256, 2) void probe(const unsigned *src, unsigned *out, int lds_bytes, int off, int spin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned *w = reinterpret_cast<unsigned *>(smem);
    for (int i = tid; i < lds_bytes / 4; i += 256) w[i] = 0xAAAAAAAAu;
    __syncthreads();
    if (wave == 0) {
        const unsigned char *s = reinterpret_cast<const unsigned char *>(src) + (size_t)blockIdx.x * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((glob_t *)s, (lds_t *)(smem + off), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    int bad_in = 0, bad_out = 0;
    for (int i = tid; i < lds_bytes / 4; i += 256) {
        const int b = i * 4;
        const unsigned v = w[i];
        if (b >= off && b < off + 1024) bad_in += v != src[(size_t)blockIdx.x * 256 + (b - off) / 4];
        else bad_out += v != 0xAAAAAAAAu;
    }
    if (bad_in) atomicAdd(&out[blockIdx.x * 4 + 0], (unsigned)bad_in);
    if (bad_out) atomicAdd(&out[blockIdx.x * 4 + 1], (unsigned)bad_out);
    if (tid == 0) {
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));  // HW_REG_LDS_ALLOC
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
    }
}

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = 2 * ncu;
    std::vector<unsigned> h((size_t)grid * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000000u + (unsigned)i;
    unsigned *src, *out;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, (size_t)grid * 16);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> r((size_t)grid * 4);
    const int sizes[] = {81920, 73728, 65536, 40960};
    const int offs[] = {0, 16384, 32768, 47104, 48128, 49152, 56320, 64512, 72704, 80896};
    for (int lds : sizes) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int off : offs) {
            if (off + 1024 > lds) continue;
            hipMemset(out, 0, (size_t)grid * 16);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, src, out, lds, off, 200);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(r.data(), out, (size_t)grid * 16, hipMemcpyDeviceToHost);
            int nbi = 0, nbo = 0, first = -1;
            for (int b = 0; b < grid; ++b) {
                nbi += r[b * 4] != 0;
                nbo += r[b * 4 + 1] != 0;
                if (first < 0 && (r[b * 4] || r[b * 4 + 1])) first = b;
            }
            printf("lds %6d off %6d: %s  wgs with a bad piece %4d / %d, wgs with foreign bytes %4d", lds, off, hipGetErrorString(e), nbi, grid, nbo);
            if (first >= 0) printf("   e.g. wg %d: bad_in %u bad_out %u LDS_ALLOC 0x%08x HW_ID 0x%08x", first, r[first * 4], r[first * 4 + 1], r[first * 4 + 2], r[first * 4 + 3]);
            printf("   (a good wg: LDS_ALLOC 0x%08x / 0x%08x)\n", r[2], r[(grid - 1) * 4 + 2]);
        }
    }
    return 0;
}
