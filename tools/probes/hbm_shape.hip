// Stand-alone probe (not part of libomnitok): what does the PEG kernels' ACCESS SHAPE cost on the MI355X, with the stencil taken
// away?  A token tensor [clips * 5 planes * 32 x 32 positions, 512 channels] fp32 (C3's: 335 MB) is copied x -> y by workgroups
// that, like peg3d_lds_kernel, own a 4 x 32 tile of positions and a slab of P bytes of the channels, walk the 5 planes of a clip,
// and (optionally) read the one-position halo around the tile.  P = 2048 is whole token rows; P = 128 is PEG's shape (one token
// row split over 16 workgroups).  Order: tiles of one (clip, slab) consecutive on one XCD (PEG's) or slabs of one tile adjacent.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_shape.hip -o tools/_bin/hbm_shape && tools/_bin/hbm_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int T = 5, H = 32, W = 32, D = 512;

// P4 = float4 per position in the slab; HALO: also read the ring of positions around the tile (values summed in, so the loads stay);
// DEEP: request all 5 planes before the first store (else plane by plane, one ahead -- the ring kernel's depth)
// TH x TW = the tile of positions: chosen so that every shape moves the same 16 KiB per workgroup and plane
template <int P4, int TH, int TW, bool HALO, bool DEEP>
__global__ __launch_bounds__(256) void shape_copy(const f32x4 *__restrict__ x, f32x4 *__restrict__ y, int clips, int xcd_order) {
    constexpr int NSLAB = (D / 4) / P4, PER = (H / TH) * (W / TW);
    const int ngroups = clips * NSLAB;
    int grp, tile;
    const int n = blockIdx.x;
    if (xcd_order) {  // groups dealt round-robin to the 8 XCDs, a group's tiles consecutive on its XCD
        const int xcd = n & 7, j = n >> 3;
        grp = xcd + 8 * (j / PER);
        tile = j % PER;
        if (grp >= ngroups) return;
    } else {          // slabs of one tile adjacent in launch order
        const int slab = n % NSLAB, r = n / NSLAB;
        tile = r % PER;
        grp = (r / PER) * NSLAB + slab;
    }
    const int slab = grp % NSLAB, b = grp / NSLAB;
    const int h0 = (tile / (W / TW)) * TH, w0 = (tile % (W / TW)) * TW;
    constexpr int RH = HALO ? TH + 2 : TH, RW = HALO ? TW + 2 : TW;
    constexpr int NV = RH * RW * P4, PT = (NV + 255) / 256;  // float4 per plane tile, per thread
    const int tid = threadIdx.x;
    auto src = [&](int t, int k, bool &inner, long &o) {
        const int i = tid + 256 * k;
        const int pos = i / P4, q = i % P4;
        const int r = pos / RW, c = pos % RW;
        const int hh = h0 + r - (HALO ? 1 : 0), ww = w0 + c - (HALO ? 1 : 0);
        const bool ok = i < NV && hh >= 0 && hh < H && ww >= 0 && ww < W;
        inner = ok && hh >= h0 && hh < h0 + TH && ww >= w0 && ww < w0 + TW;
        o = ((((long)b * T + t) * H + hh) * W + ww) * (D / 4) + slab * P4 + q;
        return ok;
    };
    f32x4 sink = {0, 0, 0, 0};  // every loaded vector is consumed (the halo loads must not be optimised away)
    if (DEEP) {
        f32x4 v[T][PT];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                bool in; long o;
                v[t][k] = src(t, k, in, o) ? x[o] : f32x4{0, 0, 0, 0};
            }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                bool in; long o;
                src(t, k, in, o);
                sink += v[t][k];
                if (in) y[o] = v[t][k];
            }
    } else {
        f32x4 cur[PT], nxt[PT];
#pragma unroll
        for (int k = 0; k < PT; ++k) { bool in; long o; cur[k] = src(0, k, in, o) ? x[o] : f32x4{0, 0, 0, 0}; }
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T)
#pragma unroll
                for (int k = 0; k < PT; ++k) { bool in; long o; nxt[k] = src(t + 1, k, in, o) ? x[o] : f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int k = 0; k < PT; ++k) { bool in; long o; src(t, k, in, o); sink += cur[k]; if (in) y[o] = cur[k]; }
#pragma unroll
            for (int k = 0; k < PT; ++k) cur[k] = nxt[k];
        }
    }
    if (sink.x == 12345.678f) y[0] = sink;  // never true for this input
}

template <int P4, int TH, int TW, bool HALO, bool DEEP>
static void run(const f32x4 *x, f32x4 *y, int clips, int xcd_order, const char *name) {
    constexpr int NSLAB = (D / 4) / P4, PER = (H / TH) * (W / TW);
    const int grid = clips * NSLAB * PER;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((shape_copy<P4, TH, TW, HALO, DEEP>), dim3(grid), dim3(256), 0, 0, x, y, clips, xcd_order);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((shape_copy<P4, TH, TW, HALO, DEEP>), dim3(grid), dim3(256), 0, 0, x, y, clips, xcd_order);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double bytes = 2.0 * clips * T * H * W * D * 4;
    printf("%-52s piece %4d B  %s  %s  %7.1f us  %5.2f TB/s (copy bytes)\n", name, P4 * 16, HALO ? "halo" : "    ",
           xcd_order ? "xcd-grouped" : "slab-adjacent", ms * 1e3, bytes / ms * 1e-9);
}

int main() {
    const int clips = 32;
    const size_t n4 = (size_t)clips * T * H * W * D / 4;
    f32x4 *x, *y;
    CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, n4 * 16));
    CK(hipMemset(x, 1, n4 * 16)); CK(hipMemset(y, 0, n4 * 16));
    printf("tensor %d clips x %d planes x %dx%d positions x %d ch fp32 = %.0f MB; copy = read + write once\n", clips, T, H, W, D, n4 * 16 / 1e6);
    for (int xo = 1; xo >= 0; --xo) {
        run<128, 1, 8, false, true>(x, y, clips, xo, "whole rows (1 x 8 positions), all planes up front");
        run<128, 1, 8, false, false>(x, y, clips, xo, "whole rows, one plane ahead");
        run<32, 1, 32, false, true>(x, y, clips, xo, "quarter rows (1 x 32), all planes up front");
        run<32, 1, 32, false, false>(x, y, clips, xo, "quarter rows, one plane ahead");
        run<16, 2, 32, false, true>(x, y, clips, xo, "eighth rows (2 x 32), all planes up front");
        run<8, 4, 32, false, true>(x, y, clips, xo, "PEG's slab (4 x 32), all planes up front");
        run<8, 4, 32, false, false>(x, y, clips, xo, "PEG's slab, one plane ahead");
        run<8, 4, 32, true, false>(x, y, clips, xo, "PEG's slab + halo, one plane ahead");
        run<8, 4, 32, true, true>(x, y, clips, xo, "PEG's slab + halo, all planes up front");
    }
    // the copied planes must be the input
    std::vector<unsigned> h(1024);
    CK(hipMemcpy(h.data(), y, 4096, hipMemcpyDeviceToHost));
    printf("check: y[0] = %08x (expect 01010101)\n", h[0]);
    return 0;
}
