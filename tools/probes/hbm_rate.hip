// Stand-alone probe: what does the STRUCTURE of a row pass cost?  layernorm_kernel (one row per wave, two loads per lane) moves C3's
// token tensor at ~6.9 TB/s; stats_pack (16 rows per wave in registers, output in 256-byte runs of the plane layout) at ~5.5 TB/s.
// Here the arithmetic is taken away: [163840, 512] fp32 is read row-major by waves that hold R rows each and written either row-major
// or in the plane layout's shape -- inside a 64-row block, chunk-major: [chunk of RUN bytes][row][RUN] -- so that a wave's stores form
// contiguous runs of R x RUN bytes (stats_pack: R = 16, RUN = 16 -> 256 B).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_rate.hip -o tools/_bin/hbm_rate && tools/_bin/hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int D4 = 128;  // float4 per row

template <int R, int RUN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void row_pass(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, long rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * WAVES + wave) * R;
    if (r0 >= rows) return;
    f32x4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) v[r][h] = x[(r0 + r) * D4 + lane + 64 * h];
    constexpr int RUN4 = RUN / 16;                 // float4 per run
    constexpr int PER_CHUNK = R * RUN4;            // float4 the wave writes per chunk
    const long blk = r0 / 64, rin = r0 % 64;       // 64-row block and the wave's first row inside it
#pragma unroll
    for (int s = 0; s < 2 * R; ++s) {
        const int p = s * 64 + lane;
        const int c = p / PER_CHUNK, q = p % PER_CHUNK, r = q / RUN4, o = q % RUN4;
        y[blk * 64 * D4 + ((long)c * 64 + rin + r) * RUN4 + o] = v[s / 2][s & 1];
    }
}

template <int R, int RUN, int WAVES>
static void run(const f32x4 *x, f32x4 *y, long rows, const char *name) {
    const int grid = (int)(rows / (R * WAVES));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((row_pass<R, RUN, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, x, y, rows);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((row_pass<R, RUN, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, x, y, rows);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-58s %2d rows/wave  %d waves/wg  store runs %5d B  %7.1f us  %5.2f TB/s\n", name, R, WAVES, R * RUN, ms * 1e3,
           2.0 * rows * D4 * 16 / ms * 1e-9);
}

int main() {
    const long rows = 163840;
    f32x4 *x, *y;
    CK(hipMalloc(&x, rows * D4 * 16)); CK(hipMalloc(&y, rows * D4 * 16));
    CK(hipMemset(x, 1, rows * D4 * 16));
    printf("[%ld, 512] fp32 = %.0f MB read + the same written\n", rows, rows * D4 * 16 / 1e6);
    run<1, 2048, 4>(x, y, rows, "row-major out, one row per wave (layernorm_kernel's shape)");
    run<2, 2048, 4>(x, y, rows, "row-major out");
    run<4, 2048, 4>(x, y, rows, "row-major out");
    run<16, 2048, 4>(x, y, rows, "row-major out, 16 rows per wave (stats_pack's register tile)");
    run<16, 16, 4>(x, y, rows, "plane-shaped out, stats_pack's shape");
    run<16, 64, 4>(x, y, rows, "plane-shaped out, 64-byte chunks");
    run<4, 16, 4>(x, y, rows, "plane-shaped out, 4 rows per wave");
    run<4, 64, 4>(x, y, rows, "plane-shaped out, 4 rows per wave, 64-byte chunks");
    run<8, 16, 4>(x, y, rows, "plane-shaped out, 8 rows per wave");
    run<8, 16, 8>(x, y, rows, "plane-shaped out, 8 rows per wave, 8 waves");
    run<16, 16, 2>(x, y, rows, "plane-shaped out, 16 rows per wave, 2 waves");
    return 0;
}
