// Stand-alone micro-benchmark of the plane x plane GEMM (csrc/gemm_pl.h) against the fp32-operand h2 kernel on the
// C3 GEMM shapes.  Links libomnitok.so only (no Python, no torch: starts in a second on the GPU box, so it is cheap
// to run under rocprofv3 --pmc).
//   build: tools/build_pl_bench.sh        run: tools/_bin/pl_bench [--iters N] [--shape name] [--cfg list] [--check]
// Per arm: ms per launch, fp32-equivalent TF, and the shader clock the launch ran at (s_memtime span of workgroup 0 /
// HIP-event time), so that arms are never compared at different clocks without knowing it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/omnitok.h"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                \
        }                                                                           \
    } while (0)
#define OK(x)                                                                    \
    do {                                                                         \
        int r_ = (x);                                                            \
        if (r_ != 0) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #x, r_, omnitok_last_error());     \
            exit(3);                                                             \
        }                                                                        \
    } while (0)

__global__ void fill_randn(float *x, int64_t n, unsigned seed, float scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    const float u1 = ((unsigned)(z & 0xFFFFFF) + 1.0f) / 16777217.0f, u2 = (unsigned)((z >> 24) & 0xFFFFFF) / 16777216.0f;
    x[i] = scale * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}

static float *dmalloc_f(int64_t n) {
    void *p;
    CK(hipMalloc(&p, (size_t)n * 4));
    return (float *)p;
}

struct Shape { const char *name; int N, K; bool geglu, residual; bool rowln = false; };

int main(int argc, char **argv) {
    int iters = 10;
    bool check = false;
    std::string only, cfgs = "1,2", h2dbg;
    int64_t L = 32 * 5120;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--shape")) only = argv[++i];
        else if (!strcmp(argv[i], "--cfg")) cfgs = argv[++i];
        else if (!strcmp(argv[i], "--check")) check = true;
        else if (!strcmp(argv[i], "--h2dbg")) h2dbg = argv[++i];  // ablation builds of the h2 kernel (plain shapes only)
        else if (!strcmp(argv[i], "--rows")) L = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--opt")) {  // --opt name value: omnitok_set_option
            const char *n = argv[++i];
            OK(omnitok_set_option(n, atoi(argv[++i])));
        }
    }
    printf("%s\n", omnitok_version());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    long long *dcyc;
    CK(hipMalloc((void **)&dcyc, 8));

    const Shape shapes[] = {{"q_or_out", 512, 512, false, false}, {"out_res", 512, 512, false, true}, {"qkv", 1536, 512, false, false},
                            {"ff_in", 2816, 512, true, false},   {"ff_out", 512, 1408, false, true}, {"out_res_ln", 512, 512, false, true, true}};
    const int64_t Lp = (L + 255) / 256 * 256;
    for (const Shape &sh : shapes) {
        if (!only.empty() && only != sh.name) continue;
        const int N = sh.N, K = sh.K, Np = (N + 255) / 256 * 256;
        float *x = dmalloc_f(L * K), *w = dmalloc_f((int64_t)N * K), *res = dmalloc_f(L * 512);
        fill_randn<<<(unsigned)((L * K + 255) / 256), 256, 0, st>>>(x, L * K, 1, 1.0f);
        fill_randn<<<(unsigned)(((int64_t)N * K + 255) / 256), 256, 0, st>>>(w, (int64_t)N * K, 2, 0.04f);
        fill_randn<<<(unsigned)((L * 512 + 255) / 256), 256, 0, st>>>(res, L * 512, 3, 1.0f);
        void *ap, *wp, *hp2 = nullptr;
        CK(hipMalloc(&ap, (size_t)Lp * K * 4));
        CK(hipMalloc(&wp, (size_t)Np * K * 4));
        float *asc = dmalloc_f(L), *wsc = dmalloc_f(N);
        OK(omnitok_pl_pack_rows(x, K, L, K, Lp, ap, asc, 0.0f, st));
        OK(omnitok_pl_pack_weight(w, K, N, K, Np, wp, wsc, st));
        const int Nout = sh.geglu ? N / 2 : N;
        float *c = dmalloc_f(L * (int64_t)Nout);
        void *outp = nullptr;
        if (sh.geglu || sh.rowln) CK(hipMalloc(&outp, (size_t)Lp * Nout * 4));
        float *gam = dmalloc_f(512), *bet = dmalloc_f(512);
        fill_randn<<<2, 256, 0, st>>>(gam, 512, 4, 0.3f);
        fill_randn<<<2, 256, 0, st>>>(bet, 512, 5, 0.1f);
        // old kernel operands
        void *wp_h2;
        CK(hipMalloc(&wp_h2, (size_t)((N + 63) / 64 * 64) * K * 4));
        float *wsc_h2 = dmalloc_f(N);
        OK(omnitok_h2_pack_weight(w, K, N, K, wp_h2, wsc_h2, st));
        float *c_h2 = dmalloc_f(L * (int64_t)Nout);
        CK(hipStreamSynchronize(st));
        const double flops = 2.0 * (double)L * N * K;

        auto run_pl = [&](int cfg, bool timed) {
            omnitok_pl_gemm g;
            memset(&g, 0, sizeof(g));
            g.a = ap; g.a_scale = asc; g.w = wp; g.w_scale = wsc;
            g.M = L; g.N = N; g.K = K; g.cfg = cfg;
            g.debug_cycles = timed ? dcyc : nullptr;
            if (sh.geglu) {
                g.epilogue = 1; g.out_planes = outp; g.out_planes_k = Nout; g.out_bound = 64.0f;
            } else if (sh.rowln) {
                g.epilogue = 2; g.c = c; g.ldc = Nout; g.residual = res; g.ldr = 512;
                g.out_planes = outp; g.out_planes_k = Nout; g.out_bound = 64.0f;
                g.ln_gamma = gam; g.ln_beta = bet; g.ln_eps = 1e-5f;
            } else {
                g.epilogue = 0; g.c = c; g.ldc = Nout;
                if (sh.residual) { g.residual = res; g.ldr = 512; }
            }
            OK(omnitok_gemm_pl(&g, st));
        };
        auto run_h2 = [&]() {
            OK(omnitok_gemm_h2(x, K, wp_h2, wsc_h2, nullptr, sh.residual ? res : nullptr, 512, c_h2, Nout, L, N, K,
                               sh.geglu ? OMNITOK_GEMM_GEGLU : (sh.residual ? OMNITOK_GEMM_RESIDUAL : 0), 0, 0, 0, 8.0f, nullptr, 1, 0,
                               nullptr, nullptr, nullptr, 0, 0.0f, nullptr, 0, 0, st));
        };
        auto time_it = [&](auto &&fn) {
            for (int i = 0; i < 3; ++i) fn();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) fn();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return ms / iters;
        };
        printf("== %s  M=%lld N=%d K=%d%s%s\n", sh.name, (long long)L, N, K, sh.geglu ? " geglu->planes" : "", sh.residual ? " +residual" : "");
        {
            const float ms = time_it(run_h2);
            printf("   h2 (fp32 A, in-loop split)     %.4f ms  %6.1f TF\n", ms, flops / ms / 1e9);
        }
        if (!sh.geglu && !sh.residual && !sh.rowln) {
            for (size_t pos = 0; pos < h2dbg.size();) {
                const int dbg = atoi(h2dbg.c_str() + pos);
                size_t nx = h2dbg.find(',', pos);
                pos = nx == std::string::npos ? h2dbg.size() : nx + 1;
                OK(omnitok_set_option("h2_dbg", dbg));
                const float ms = time_it(run_h2);
                OK(omnitok_set_option("h2_dbg", 0));
                printf("   h2 ablation build dbg=%-2d        %.4f ms  %6.1f TF\n", dbg, ms, flops / ms / 1e9);
            }
        }
        for (size_t pos = 0; pos < cfgs.size();) {
            const int cfg = atoi(cfgs.c_str() + pos);
            size_t nx = cfgs.find(',', pos);
            pos = nx == std::string::npos ? cfgs.size() : nx + 1;
            const float ms = time_it([&] { run_pl(cfg, false); });
            // one more launch alone for the clock: span of workgroup 0 in shader cycles / its wall time
            CK(hipMemsetAsync(dcyc, 0, 8, st));
            CK(hipEventRecord(e0, st));
            run_pl(cfg, true);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms1;
            CK(hipEventElapsedTime(&ms1, e0, e1));
            long long cyc = 0;
            CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
            printf("   pl cfg %d                       %.4f ms  %6.1f TF   (wg0 span %lld cyc in %.4f ms launch)\n", cfg, ms,
                   flops / ms / 1e9, cyc, ms1);
        }
        if (check) {
            // fp64 reference on sampled rows, from the fp32 inputs
            run_pl(atoi(cfgs.c_str()), false);
            if (!sh.geglu) {
                // pl vs h2 on everything
                run_h2();
            }
            CK(hipStreamSynchronize(st));
            std::vector<float> hw((size_t)N * K), hx((size_t)K), hr(512);
            CK(hipMemcpy(hw.data(), w, hw.size() * 4, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0;
            const int64_t rows[] = {0, 1, 31, 32, 63, 64, 127, 128, 255, 256, 1000, L / 2 + 17, L - 257, L - 1};
            for (int64_t m : rows) {
                CK(hipMemcpy(hx.data(), x + m * K, (size_t)K * 4, hipMemcpyDeviceToHost));
                if (sh.residual) CK(hipMemcpy(hr.data(), res + m * 512, 512 * 4, hipMemcpyDeviceToHost));
                std::vector<double> ref(N);
                for (int n = 0; n < N; ++n) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)hx[k] * hw[(size_t)n * K + k];
                    ref[n] = s;
                }
                if (sh.geglu) {
                    // hidden planes -> value
                    std::vector<uint16_t> blk(4096);
                    for (int j = 0; j < Nout; ++j) {
                        // packed layout: 64-column groups [32 value | 32 gate]
                        const int grp = j / 32, jj = j % 32;
                        const double val = ref[grp * 64 + jj], gate = ref[grp * 64 + 32 + jj];
                        const double hid = 0.5 * gate * (1.0 + erf(gate / sqrt(2.0))) * val;
                        // read hi / lo
                        const int64_t off = (((m >> 6) * (Nout >> 5) + (j >> 5)) * 8 + ((j >> 3) & 3)) * 512 + (m & 63) * 8 + (j & 7);
                        uint16_t h, l;
                        CK(hipMemcpy(&h, (uint16_t *)outp + off, 2, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(&l, (uint16_t *)outp + off + 4 * 512, 2, hipMemcpyDeviceToHost));
                        auto f16 = [](uint16_t v) {
                            const int s = v >> 15, e = (v >> 10) & 31, f = v & 1023;
                            double r = e == 0 ? ldexp((double)f, -24) : ldexp((double)(f + 1024), e - 25);
                            return s ? -r : r;
                        };
                        const double got = (f16(h) + f16(l)) / 256.0;  // out_bound 64 = 0.5 * 2^7 -> scale 2^(15 - 7)
                        maxerr = fmax(maxerr, fabs(got - hid));
                        maxref = fmax(maxref, fabs(hid));
                        if (j >= 96 && m > 2) break;  // element-wise copies are slow: sample
                    }
                } else {
                    std::vector<float> got(N);
                    CK(hipMemcpy(got.data(), c + m * N, (size_t)N * 4, hipMemcpyDeviceToHost));
                    std::vector<double> xo(N);
                    for (int n = 0; n < N; ++n) {
                        const double r = ref[n] + (sh.residual ? hr[n] : 0.0);
                        xo[n] = r;
                        maxerr = fmax(maxerr, fabs(got[n] - r));
                        maxref = fmax(maxref, fabs(r));
                    }
                    if (sh.rowln) {
                        std::vector<float> hg(512), hb(512);
                        CK(hipMemcpy(hg.data(), gam, 2048, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(hb.data(), bet, 2048, hipMemcpyDeviceToHost));
                        double mean = 0, var = 0;
                        for (int n = 0; n < N; ++n) mean += xo[n];
                        mean /= N;
                        for (int n = 0; n < N; ++n) var += (xo[n] - mean) * (xo[n] - mean);
                        var /= N;
                        std::vector<uint16_t> rowp((size_t)N * 2);
                        for (int n = 0; n < N; ++n) {
                            const double y = (xo[n] - mean) / sqrt(var + 1e-5) * hg[n] + hb[n];
                            const int64_t off = (((m >> 6) * (N >> 5) + (n >> 5)) * 8 + ((n >> 3) & 3)) * 512 + (m & 63) * 8 + (n & 7);
                            uint16_t h, l;
                            CK(hipMemcpy(&h, (uint16_t *)outp + off, 2, hipMemcpyDeviceToHost));
                            CK(hipMemcpy(&l, (uint16_t *)outp + off + 4 * 512, 2, hipMemcpyDeviceToHost));
                            auto f16 = [](uint16_t v) {
                                const int sg = v >> 15, e = (v >> 10) & 31, f = v & 1023;
                                double r = e == 0 ? ldexp((double)f, -24) : ldexp((double)(f + 1024), e - 25);
                                return sg ? -r : r;
                            };
                            const double got2 = (f16(h) + f16(l)) / 256.0;
                            maxerr = fmax(maxerr, fabs(got2 - y));
                            if (n >= 64 && m > 2) break;
                        }
                    }
                }
            }
            printf("   check: max |pl - fp64| = %.3e (max |ref| %.3e)\n", maxerr, maxref);
            if (!sh.geglu) {
                std::vector<float> a((size_t)L * N), b((size_t)L * N);
                CK(hipMemcpy(a.data(), c, a.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), c_h2, b.size() * 4, hipMemcpyDeviceToHost));
                double md = 0;
                for (size_t i = 0; i < a.size(); ++i) md = fmax(md, fabs((double)a[i] - b[i]));
                printf("   check: max |pl - h2| over all %lld outputs = %.3e\n", (long long)a.size(), md);
            }
            if (!(maxerr < 1e-4 * fmax(1.0, maxref))) {
                printf("   CHECK FAILED\n");
                return 1;
            }
        }
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(res)); CK(hipFree(ap)); CK(hipFree(wp)); CK(hipFree(asc)); CK(hipFree(wsc));
        CK(hipFree(c)); CK(hipFree(wp_h2)); CK(hipFree(wsc_h2)); CK(hipFree(c_h2));
        if (outp) CK(hipFree(outp));
        (void)hp2;
    }
    return 0;
}
