// The fused temporal stage's own GEMM kernel (T' == 5; reference attention.py:402-486 with is_spatial = False): the same
// plane x plane products and the same two epilogues as gemm_pl.h's PL_TSCORE / PL_TPV, in a K loop built for TWO waves per
// SIMD.  The first fused form ran its 320 x 128 tiles on PlCfg<2, 2, 4, 2, 0, 2, 5> -- one wave per SIMD, 4 x 28 KiB ring --
// and the matrix pipe sat at 42 / 37 % busy: with 10 MFMAs per product group nothing fills the barrier and fragment-read
// bubbles of a single wave (profiles/r05_temporal_fused.txt).  Here
//   * a workgroup is still 4 waves x (2 x 5 accumulator blocks) on a 320 x 128 tile, but TWO workgroups share a CU:
//     80 KiB of LDS each = a 2-stage ring (2 x 28 KiB, the DMA one K step ahead) + 24 KiB of epilogue scratch;
//   * a wave fits 256 registers: 160 accumulators, the step's 4 weight fragments, and the activation fragments STREAMED one
//     row block at a time (read for block mi + 1 while block mi multiplies) instead of two whole sets held across the step;
//   * per accumulator block the products of a K step arrive in the order W lo . A hi, W hi . A hi, W hi . A lo -- the order of
//     gemm_pl_kernel -- so the accumulators are those of the first form; the epilogues round differently in the last bit (scalar
//     arithmetic here, see the two rules below) and agree with it to 8e-7 on the softmax weights.
// One workgroup's epilogue (VALU, LDS exchange, stores) now overlaps the other's K loop on the same CU: launches of 533 / 301 us
// against 614 / 329 us, -0.9 ms per step (profiles/r05_temporal_plt.txt).
#pragma once
#include "gemm_pl.h"

namespace omnitok {

constexpr int PLT_TM = 320, PLT_TN = 128, PLT_STAGE = (PLT_TM + PLT_TN) * 64, PLT_RING = 2 * PLT_STAGE;
constexpr int PLT_SCRATCH = 24576, PLT_LDS = PLT_RING + PLT_SCRATCH;  // 81920: two workgroups per CU
constexpr int PLT_PPW = 7;                                             // (4 * 5 + 4 * 2) pieces of 1 KiB per K step / 4 waves

// The 8 values of run c of an accumulator block are final HERE: nothing that produces them may sink below this point (the DAG
// scheduler otherwise floats every run's arithmetic below all of the runs' lane broadcasts: 64 live registers the wave does not have)
__device__ __forceinline__ void plt_pin8(f32x16 &a, int o) {
    float t0 = a[o], t1 = a[o + 1], t2 = a[o + 2], t3 = a[o + 3], t4 = a[o + 4], t5 = a[o + 5], t6 = a[o + 6], t7 = a[o + 7];
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7));
    a[o] = t0, a[o + 1] = t1, a[o + 2] = t2, a[o + 3] = t3, a[o + 4] = t4, a[o + 5] = t5, a[o + 6] = t6, a[o + 7] = t7;
}

// TWO RULES OF THIS FILE, both measured (profiles/r05_temporal_plt.txt):
//  1. No packed-fp32 VALU instructions.  gemm_plt.hip is built with -fno-slp-vectorize (build.py), and nothing here uses 2-element
//     float vector arithmetic.  Every build in which SLP vectorisation turned the P . V arithmetic into v_pk_mul_f32 / v_pk_fma_f32
//     (op_sel broadcasts of the softmax weight) returned the LOW half of four consecutive packed FMAs as if the product were zero,
//     in the last 16 lanes of ~2 % of the waves, 30 runs of 30 -- and only with two workgroups per CU (one per CU: bit-identical to
//     the scalar build).  The same source built without SLP, with or without the rest of this list, passed 90 runs of 90.  tests/
//     test_build_invariants.py fails the build if a v_pk_*_f32 shows up in this kernel's ISA.
//  2. No spills.  The wave has exactly 256 registers (two waves per SIMD), 160 of them accumulators; what keeps the epilogues inside
//     that: lane coordinates re-derived from `lane16` where they are needed instead of kept across phases, column constants
//     broadcast one 16-column run at a time (plt_pin8), the softmax weights of launch 2 and the learned scales of launch 1 requested
//     late, 32-bit offsets from scalar bases, the two roles of launch 1 in separate code between their three barriers.  The same
//     test fails the build on a scratch instruction in this kernel.
// v = acc * s + t with s a power of two (operand scale x weight scale): acc * s is exact, so multiply-then-add -- two 2-operand
// instructions, each in the accumulator's own register -- rounds exactly like the fused form (whose v_fmac would put the result in
// t's register and rebuild the 16-register accumulator block out of place).
__device__ __forceinline__ float plt_scale_add(float a, float s, float t) {
#pragma clang fp contract(off)
    a = a * s;
    return a + t;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_plt_kernel(PlParams p) {
    static_assert(EPI == PL_TSCORE || EPI == PL_TPV, "temporal epilogues only");
    constexpr int TM = PLT_TM, TN = PLT_TN, MI = 5, NI = 2, NW = 4, PPW = PLT_PPW, STAGE = PLT_STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32_ = lane & 31, hi_ = lane >> 5;
    const int wn_ = wave & 1, wm_ = wave >> 1;
    if ((int)blockIdx.x >= p.ntiles) return;
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nk = p.nk, kblocks = p.K >> 5;
    const int total = my_tiles * nk;

    // ---- DMA pieces of this wave: piece q = wave + 4 j; pieces 0..19 are the activation panel, 20..27 the weight panel ----
    unsigned pv_off[PPW];
    int pl_dst[PPW];
    bool p_is_w[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = wave + NW * j;
        const bool is_w = q >= 4 * (TM / 64);
        const int qq = is_w ? q - 4 * (TM / 64) : q;
        const int rowsb = is_w ? TN / 64 : TM / 64;
        const int plkg = qq / rowsb, rb = qq % rowsb;
        p_is_w[j] = is_w;
        pv_off[j] = (unsigned)rb * (unsigned)kblocks * 8192u + (unsigned)(plkg >> 1) * 4096u + (unsigned)(plkg & 1) * 1024u;
        pl_dst[j] = (is_w ? TM * 64 : 0) + qq * 1024;
    }
    const unsigned lane16 = (unsigned)lane * 16u;  // the only per-lane part of a piece address: scalar base + 32-bit lane offset
    const unsigned char *d_a = nullptr, *d_w = nullptr;
    int d_tile = -1, d_k = 0;
    auto d_set_tile = [&](int ti) {
        const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
        int64_t bm;
        int bn;
        tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);
        d_a = p.a + ((bm * TM) >> 6) * (int64_t)kblocks * 8192;
        d_w = p.w + (int64_t)bn * (TN / 64) * (int64_t)kblocks * 8192;
        d_tile = ti;
        d_k = 0;
    };
    auto dma_step = [&](int stage) {  // the pieces of K step d_k of tile d_tile -> ring stage `stage`; advances the cursor
        const unsigned koff = (unsigned)(d_k >> 1) * 8192u + (unsigned)(d_k & 1) * 2048u;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const unsigned char *src = (p_is_w[j] ? d_w : d_a) + (koff + pv_off[j]);
            __builtin_amdgcn_global_load_lds((pl_glob_t *)(src + lane16), (pl_lds_t *)(pl_smem + stage * STAGE + pl_dst[j]), 16, 0, 0);
        }
        if (++d_k == nk && d_tile + 1 < my_tiles) d_set_tile(d_tile + 1);
    };

    f32x16 acc[NI][MI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.0f;
    };
    zero_acc();
    int a_fr = 0, w_fr = 0;  // fragment bases of this lane, rebuilt from lane16 at the top of every tile (see set_fr)
    auto set_fr = [&]() {
        unsigned l = lane16;
        asm volatile("" : "+v"(l));
        const int r = (int)(l >> 4) & 31, h = (int)(l >> 9);
        a_fr = (h * TM + wm_ * (32 * MI) + r) * 16;                // + plane * (2 * TM * 16) + mi * 512
        w_fr = TM * 64 + (h * TN + wn_ * (32 * NI) + r) * 16;      // + plane * (2 * TN * 16) + ni * 512
    };
    auto rdA = [&](int stage, int plane, int mi) {
        return *reinterpret_cast<const u32x4 *>(pl_smem + stage * STAGE + a_fr + plane * (2 * TM * 16) + mi * 512);
    };
    auto rdW = [&](int stage, int plane, int ni) {
        return *reinterpret_cast<const u32x4 *>(pl_smem + stage * STAGE + w_fr + plane * (2 * TN * 16) + ni * 512);
    };
    auto mfma = [&](const u32x4 &wf, const u32x4 &af, f32x16 &c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, af), c, 0, 0, 0);
    };

    d_set_tile(0);
    dma_step(0);
    for (int ti = 0; ti < my_tiles; ++ti) {
        set_fr();
        for (int k = 0; k < nk; ++k) {
            const int s = ti * nk + k, st = s & 1;
            // this wave's pieces of step s have landed; after the barrier so have everyone's, and nobody still reads stage
            // st ^ 1 (step s - 1, or the previous tile's epilogue blocks) -- which the DMA of step s + 1 now refills
            PL_WAIT_VM(0);
            lds_barrier();
            if (s + 1 < total) dma_step(st ^ 1);
            u32x4 Ph[NI], Pl[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                Ph[ni] = rdW(st, 0, ni);
                Pl[ni] = rdW(st, 1, ni);
            }
            u32x4 X = rdA(st, 0, 0), Y = rdA(st, 1, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                u32x4 Xn = X, Yn = Y;
                if (mi + 1 < MI) {  // the next row block's fragments are in flight while this one multiplies
                    Xn = rdA(st, 0, mi + 1);
                    Yn = rdA(st, 1, mi + 1);
                }
                mfma(Pl[0], X, acc[0][mi]);
                mfma(Pl[1], X, acc[1][mi]);
                mfma(Ph[0], X, acc[0][mi]);
                mfma(Ph[1], X, acc[1][mi]);
                mfma(Ph[0], Y, acc[0][mi]);
                mfma(Ph[1], Y, acc[1][mi]);
                X = Xn;
                Y = Yn;
            }
        }
        // ---- epilogue of tile ti ---------------------------------------------------------------------------------------
        {
            const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
            int64_t bm;
            int bn;
            tile_coords(lid, p.nbm, p.nbn, p.gn, bm, bn);
            // the lane coordinates are re-derived from the one per-lane value the K loop keeps anyway (and laundered: the epilogue's
            // address arithmetic must not be hoisted above the K loop -- every register there is spoken for)
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            const int r32 = (int)(l16 >> 4) & 31, hi = (int)(l16 >> 9);
            int wv = wave;  // likewise the wave's role: constants selected by it are built per tile, not held in VGPRs across the K loop
            asm volatile("" : "+s"(wv));
            const int wn = wv & 1, wm = wv >> 1;
            const int n_w0 = bn * TN + wn * (32 * NI);
            const int64_t m_w0 = bm * TM + wm * (32 * MI);
            // the ring stage the last K step read is free until the next step's barrier; with the scratch block behind the ring
            // that is one 24 KiB region per sequence half (exchange) / 10 KiB per wave (plane staging)
            const int st_last = (ti * nk + nk - 1) & 1;
            unsigned char *free_stage = pl_smem + st_last * STAGE, *scratch = pl_smem + PLT_RING;
            lds_barrier();  // every wave is done with the last step's fragments before the blocks below are written
            // row factors and column constants (unconditional loads from clamped addresses; the other workgroup of this CU and
            // the other wave of this SIMD cover the round trip)
            float sa[MI], fmu[MI], frs[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const unsigned m = (unsigned)min((int64_t)(m_w0 + mi * 32 + r32), p.M - 1);  // rows < 2^31 (launcher)
                sa[mi] = p.a_scale[m] * p.a_scale_const;
                const float2 ms = reinterpret_cast<const float2 *>(p.fold_stats)[m];
                fmu[mi] = ms.x;
                frs[mi] = ms.y;
            }
            const int ln = hi * 32 + r32, ncl = n_w0 < p.N ? n_w0 + ln : 0;
            float swl = p.w_scale[ncl];
            const int seq = (int)(bm * 64) + wm * 32 + r32;
            const bool live = seq < p.t_nseq;
            const int seqc = live ? seq : 0;
            if constexpr (EPI == PL_TSCORE) {
                const bool isq = wn == 0;  // wave-uniform: q_h wave or k_h wave of its sequence half
                const int head = n_w0 >> 7;
                const float *fxp = isq ? p.fold_b : p.fold_u;
                const float *fxb = fxp ? fxp : p.w_scale;
                float fxl = fxb[ncl] * (fxp ? 1.0f : 0.0f);
                // v = r_a (acc sa sw) + r_b f with (r_a, r_b) = (rstd, 1) for q and (1, mean) for k; sa and sw are powers of two, so
                // acc sw is exact and (r_a sa) is exact: (acc sw)(r_a sa) rounds once, like r_a (acc (sa sw)) -- two row factors kept, not three
                float ss[MI], rsa[MI], rb[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    ss[mi] = 0.0f;
                    rsa[mi] = (isq ? frs[mi] : 1.0f) * sa[mi];
                    rb[mi] = isq ? 1.0f : fmu[mi];
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" : "+v"(swl), "+v"(fxl));  // this run's 16 lane broadcasts start HERE (256 registers: not all 64 up front)
                        float sw[8], fx[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int src = (ni * 32 + c * 16 + hi * 8 + e) * 4;
                            sw[e] = lane_bcast(swl, src);
                            fx[e] = lane_bcast(fxl, src);
                        }
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    v[e] = (acc[ni][mi][c * 8 + h * 4 + e] * sw[h * 4 + e]) * rsa[mi] + rb[mi] * fx[h * 4 + e];
                                ss[mi] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[ni][mi][c * 8 + h * 4 + e] = v[e];
                            }
                            plt_pin8(acc[ni][mi], c * 8);
                        }
                        asm volatile("" : "+v"(ss[0]), "+v"(ss[1]), "+v"(ss[2]), "+v"(ss[3]), "+v"(ss[4]));
                    }
                // the learned per-channel scale of this lane's column is requested only now (its register is not free earlier)
                unsigned l16s = lane16;
                asm volatile("" : "+v"(l16s));
                float svl = (isq ? p.q_scale : p.k_scale)[l16s >> 4];
                float inv[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) inv[mi] = 1.0f / fmaxf(sqrtf(halves_sum(ss[mi])), 1e-12f);
                const float mul = isq ? p.t_scale : 1.0f;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" : "+v"(svl));
                        float s8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) s8[e] = lane_bcast(svl, (ni * 32 + c * 16 + hi * 8 + e) * 4);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[ni][mi][c * 8 + e] = acc[ni][mi][c * 8 + e] * inv[mi] * s8[e] * mul;
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) plt_pin8(acc[ni][mi], c * 8);
                    }
                // the k wave of a half hands its k to the q wave of the same half in two rounds (steps 0..2, then 3..4)
                unsigned char *xb = wm ? scratch : free_stage;
                // block of a time step: [16 chunks of 16 B (ni, c, hi, h)][32 sequences]: the 32 lanes of an instruction touch 512
                // consecutive bytes (no bank conflicts either way) and every chunk is lane base + a compile-time offset
                // (lane coordinates derived again: what the projection phases above keep alive must not include the addresses below)
                unsigned l16x = lane16;
                asm volatile("" : "+v"(l16x));
                const int r32x = (int)(l16x >> 4) & 31, hix = (int)(l16x >> 9);
                unsigned char *xl = xb + hix * 1024 + r32x * 16;
                const int seqx = (int)(bm * 64) + wm * 32 + r32x;
                const bool livex = seqx < p.t_nseq;
                float sc[15];
                // softmax rows [i0, i1) of this lane's (sequence, head): complete as soon as key step i1 - 1 has been read, so rows
                // 0..2 leave after round 0 (12 fewer values live across the second round)
                auto emit_rows = [&](int i0, int i1) {
                    if (n_w0 >= p.N) return;
                    const float slope = p.t_alibi ? p.t_alibi[head] : 0.0f;
                    float *pp = p.tp + (unsigned)(((livex ? seqx : 0) * p.t_heads + head) * 40);
                    const bool wr = hix == 0 && livex;
#pragma unroll
                    for (int i = i0; i < i1; ++i) {
                        float mx = -INFINITY, dd[5];
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            dd[j] = sc[i * (i + 1) / 2 + j] - slope * (float)(i - j);
                            mx = fmaxf(mx, dd[j]);
                        }
                        float l = 0.0f, ev[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int j = 0; j <= i; ++j) {
                            ev[j] = expf(dd[j] - mx);
                            l += ev[j];
                        }
                        if (wr) {
                            *reinterpret_cast<f32x4 *>(pp + i * 8) = f32x4{ev[0], ev[1], ev[2], ev[3]};
                            *reinterpret_cast<f32x4 *>(pp + i * 8 + 4) = f32x4{ev[4], 1.0f / l, 0.0f, 0.0f};
                        }
                    }
                };
                // The two roles run their own code between the three barriers both execute (B1: round 0 written, B2: round 0 read,
                // B3: round 1 written): a score's registers then never cross the other role's code.
                auto put_round = [&](int s0, int s1) {
#pragma unroll
                    for (int s = s0; s < s1; ++s)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int c = 0; c < 2; ++c)
#pragma unroll
                                for (int h = 0; h < 2; ++h)
                                    *reinterpret_cast<f32x4 *>(xl + (s - s0) * 8192 + (ni * 8 + c * 4 + h) * 512) =
                                        f32x4{acc[ni][s][c * 8 + h * 4], acc[ni][s][c * 8 + h * 4 + 1], acc[ni][s][c * 8 + h * 4 + 2],
                                              acc[ni][s][c * 8 + h * 4 + 3]};
                };
                auto dot_round = [&](int s0, int s1) {
#pragma unroll
                    for (int s = s0; s < s1; ++s) {
                        // the chain of a score runs over the 64 channels in the order (ni, c, h, e); k arrives one 32-channel block at
                        // a time (16 registers instead of 32: the wave has 256)
                        float d[5];
#pragma unroll
                        for (int t = 0; t < 5; ++t) d[t] = 0.0f;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            f32x4 kk[4];
#pragma unroll
                            for (int c = 0; c < 2; ++c)
#pragma unroll
                                for (int h = 0; h < 2; ++h)
                                    kk[c * 2 + h] = *reinterpret_cast<const f32x4 *>(xl + (s - s0) * 8192 + (ni * 8 + c * 4 + h) * 512);
#pragma unroll
                            for (int t = s; t < 5; ++t)
#pragma unroll
                                for (int c = 0; c < 2; ++c)
#pragma unroll
                                    for (int h = 0; h < 2; ++h)
#pragma unroll
                                        for (int e = 0; e < 4; ++e) d[t] = __builtin_fmaf(acc[ni][t][c * 8 + h * 4 + e], kk[c * 2 + h][e], d[t]);
                        }
#pragma unroll
                        for (int t = s; t < 5; ++t) sc[t * (t + 1) / 2 + s] = halves_sum(d[t]);  // v_permlane32_swap: no LDS crossbar trip
                    }
                };
                if (!isq) {
                    put_round(0, 3);
                    lds_barrier();
                    lds_barrier();
                    put_round(3, 5);
                    lds_barrier();
                } else {
                    lds_barrier();
                    dot_round(0, 3);
                    emit_rows(0, 3);
                    lds_barrier();
                    lds_barrier();
                    dot_round(3, 5);
                    emit_rows(3, 5);
                }
            } else {
                const float *fub = p.fold_u ? p.fold_u : p.w_scale;
                float ful = fub[ncl] * (p.fold_u ? 1.0f : 0.0f);
                const float *vbp = p.v_bound_dev ? p.v_bound_dev + (seqc / p.t_seqs_per_clip) * p.v_bound_stride : p.w_scale;
                const float vbd = p.v_bound_dev ? *vbp : 1.0f + 0.0f * *vbp;
                const float so = h2_scale_of_bound(p.v_bound * vbd);
                const float *pp = p.tp + (unsigned)((seqc * p.t_heads + (n_w0 < p.N ? n_w0 >> 6 : 0)) * 40);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" : "+v"(swl), "+v"(ful));
                        float sw[8], fx[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int src = (ni * 32 + c * 16 + hi * 8 + e) * 4;
                            sw[e] = lane_bcast(swl, src);
                            fx[e] = lane_bcast(ful, src);
                        }
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                acc[ni][mi][c * 8 + e] = plt_scale_add(acc[ni][mi][c * 8 + e], sa[mi] * sw[e], fmu[mi] * fx[e]);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) plt_pin8(acc[ni][mi], c * 8);
                    }
                // the softmax weights of launch 1 are requested only now: the 40 registers are free once the row factors and the
                // broadcast column constants above are dead (the other wave of this SIMD multiplies during the round trip)
                asm volatile("" : "+v"(pp));
                f32x4 pw[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) pw[q] = *reinterpret_cast<const f32x4 *>(pp + q * 4);
#pragma unroll
                for (int i = 4; i >= 0; --i) {
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 pa = pw[2 * i], pb = pw[2 * i + 1];
                    const float e0 = pa[0], e1 = pa[1], e2 = pa[2], e3 = pa[3], e4 = pb[0];
                    const float fin = pb[1] * so;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float o = acc[ni][0][r] * e0;
                            if (i >= 1) o = fmaf(acc[ni][1][r], e1, o);
                            if (i >= 2) o = fmaf(acc[ni][2][r], e2, o);
                            if (i >= 3) o = fmaf(acc[ni][3][r], e3, o);
                            if (i >= 4) o = fmaf(acc[ni][4][r], e4, o);
                            acc[ni][i][r] = o * fin;
                        }
                }
                if (n_w0 < p.N) {
                    unsigned char *stg = (wm ? scratch : free_stage) + wn * 10240;
                    const int row0 = ((int)(bm * 64) + wm * 32) * 5;   // token rows and plane bytes fit 32 bits (launcher)
                    const int lane64 = hi * 32 + r32;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < 5; ++i) {
                                f32x4 va, vb2;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    va[e] = acc[ni][i][c * 8 + e];
                                    vb2[e] = acc[ni][i][c * 8 + 4 + e];
                                }
                                const f16x4 ha = __builtin_convertvector(va, f16x4), hb = __builtin_convertvector(vb2, f16x4);
                                const f16x4 la = __builtin_convertvector(va - __builtin_convertvector(ha, f32x4), f16x4);
                                const f16x4 lb = __builtin_convertvector(vb2 - __builtin_convertvector(hb, f32x4), f16x4);
                                unsigned char *d = stg + (hi * 160 + r32 * 5 + i) * 16;
                                *reinterpret_cast<u32x4 *>(d) = __builtin_bit_cast(u32x4, __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7));
                                *reinterpret_cast<u32x4 *>(d + 5120) = __builtin_bit_cast(u32x4, __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7));
                            }
                            __builtin_amdgcn_s_waitcnt(0xC07F);
                            __builtin_amdgcn_wave_barrier();
                            const int kb = (n_w0 >> 5) + ni;
#pragma unroll
                            for (int pl2 = 0; pl2 < 2; ++pl2)
#pragma unroll
                                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                                    for (int j = 0; j < 3; ++j) {
                                        const int rr = j * 64 + lane64;
                                        const u32x4 w = *reinterpret_cast<const u32x4 *>(stg + ((pl2 * 2 + hh) * 160 + (rr < 160 ? rr : 0)) * 16);
                                        const int m = row0 + rr;
                                        const unsigned off = (unsigned)((m >> 6) * p.cp_kblocks + kb) * 8192u + (unsigned)(pl2 * 4096 + (c * 2 + hh) * 1024) +
                                                             (unsigned)(m & 63) * 16u;
                                        if (rr < 160 && m < p.t_nseq * 5) *reinterpret_cast<u32x4 *>(p.cp + off) = w;
                                    }
                            __builtin_amdgcn_wave_barrier();
                        }
                    if (live && n_w0 == 0 && hi == 0) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) p.t_out_scale[(unsigned)(seqc * 5 + i)] = 1.0f / so;
                    }
                }
            }
        }
        zero_acc();
    }
}

}  // namespace omnitok
