// Stand-alone probe: omnitok::pegw::peg3d_wide_kernel (omnitokenizer_amd/csrc/peg_wide.h) against the in-tree PEG kernel called through the C ABI
// (libomnitok.so): bit equality and time per launch at the C3 shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -Iinclude tools/probes/peg_wide_probe.hip -Lomnitokenizer_amd/lib -lomnitok \
//         -Wl,-rpath,'$ORIGIN/../../omnitokenizer_amd/lib' -o tools/_bin/peg_wide_probe
#include "../../omnitokenizer_amd/csrc/peg_wide.h"
#include "omnitok.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void launch_wide(const float *x, const float *w27, const float *bias, float *y, int B, int T, int H, int W, int D, int causal) {
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(omnitok::pegw::peg3d_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, omnitok::pegw::LDS_BYTES));
        once = true;
    }
    const int grid = (W / omnitok::pegw::W_T) * (H / omnitok::pegw::H_T) * (D / 64) * B;
    hipLaunchKernelGGL(omnitok::pegw::peg3d_wide_kernel, dim3(grid), dim3(256), omnitok::pegw::LDS_BYTES, 0, x, w27, bias, y, B, T, H, W, D, causal ? 2 : 1);
}

static float time_ms(void (*fn)(void *), void *ctx, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn(ctx);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) fn(ctx);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct Ctx { const float *x, *w27, *bias; float *y; int B, T, H, W, D, causal; };

int main() {
    const int D = 512;
    const int shapes[][5] = {{32, 5, 32, 32, 1}, {32, 5, 32, 32, 0}, {160, 1, 32, 32, 1}, {2, 7, 8, 16, 0}, {1, 17, 64, 64, 1}, {3, 2, 4, 16, 1}};
    for (auto &s : shapes) {
        const int B = s[0], T = s[1], H = s[2], W = s[3], causal = s[4];
        const size_t n = (size_t)B * T * H * W * D;
        std::vector<float> hx(n), hw(27 * D), hb(D);
        unsigned r = 12345u + (unsigned)n;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; return ((r >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto &v : hx) v = rnd();
        for (auto &v : hw) v = rnd() * 0.2f;
        for (auto &v : hb) v = rnd() * 0.05f;
        float *x, *w27, *wraw, *bias, *y0, *y1;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y0, n * 4)); CK(hipMalloc(&y1, n * 4));
        CK(hipMalloc(&w27, 27 * D * 4)); CK(hipMalloc(&wraw, 27 * D * 4)); CK(hipMalloc(&bias, D * 4));
        CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(wraw, hw.data(), 27 * D * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), D * 4, hipMemcpyHostToDevice));
        if (omnitok_pack_peg_weight(wraw, D, w27, nullptr)) { printf("pack: %s\n", omnitok_last_error()); return 1; }
        CK(hipMemset(y0, 0xFF, n * 4)); CK(hipMemset(y1, 0xEE, n * 4));
        if (omnitok_peg3d(x, w27, bias, y0, B, T, H, W, D, causal, nullptr)) { printf("peg3d: %s\n", omnitok_last_error()); return 1; }
        launch_wide(x, w27, bias, y1, B, T, H, W, D, causal);
        CK(hipDeviceSynchronize());
        std::vector<float> a(n), b(n);
        CK(hipMemcpy(a.data(), y0, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), y1, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
        Ctx c{x, w27, bias, y0, B, T, H, W, D, causal};
        const float t_ref = time_ms([](void *p) { Ctx *c = (Ctx *)p; omnitok_peg3d(c->x, c->w27, c->bias, c->y, c->B, c->T, c->H, c->W, c->D, c->causal, nullptr); }, &c, 20);
        c.y = y1;
        const float t_new = time_ms([](void *p) { Ctx *c = (Ctx *)p; launch_wide(c->x, c->w27, c->bias, c->y, c->B, c->T, c->H, c->W, c->D, c->causal); }, &c, 20);
        const double gb = 2.0 * n * 4 / 1e9;
        printf("[%3d x %2d x %2dx%2d %s] differing floats %zu of %zu | in-tree %7.1f us %.2f TB/s | 64-ch slab %7.1f us %.2f TB/s\n", B, T, H, W,
               causal ? "causal    " : "non-causal", bad, n, t_ref * 1e3, gb / t_ref, t_new * 1e3, gb / t_new);
        CK(hipFree(x)); CK(hipFree(y0)); CK(hipFree(y1)); CK(hipFree(w27)); CK(hipFree(wraw)); CK(hipFree(bias));
    }
    return 0;
}
