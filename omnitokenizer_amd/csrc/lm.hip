// LM consumer of the token path: KV-cached decode step of the reference's minGPT
// (OmniTokenizer/modules/gpt.py:74-275) for sample_with_past (gpt.py:327-359).
//
// One token per stream per step, so every matrix product is a GEMV over fp32 weights: the step is
// HBM-bound (24 layers x 12 C^2 x 4 B = 2.7 GB of weights at C = 1536, plus the K/V cache), not
// MFMA-bound.  Kernels:
//   lm_embed_kernel        x = tok_emb[idx] + pos_emb[pos]
//   lm_gemv_kernel         y = act(LN(x) W^T + b) (+ residual): one wave streams ROWS weight rows with
//                          16-byte loads (each row read exactly once for up to 8 batch rows), LayerNorm
//                          of the (tiny, L2-resident) activations fused into the prologue, bias / exact
//                          GELU / residual fused into the epilogue
//   lm_attn_decode_kernel  flash-decode over the K/V cache: 256-key chunks per workgroup, 8 lanes per
//                          key (coalesced 128-byte row segments), online softmax, partial (m, l, o)
//   lm_attn_merge_kernel   merges the chunk partials
//   lm_advance_kernel      cache_len++, pos++ on the device
// Shapes and launch grids do not depend on the position (chunks beyond the cache length exit at
// once), so a step can be captured once in a HIP graph and replayed for every token.
#include "common.h"
#include "gemm_x_common.h"  // gelu_erf, x3_rsrc
#include "../../include/omnitok_lm.h"

#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>

namespace omnitok {

constexpr int LM_CHUNK = 256;   // keys per attention workgroup
constexpr int LM_CHUNK_SHORT = 128;  // ... of the decode step when the cache holds at most 32 such chunks: twice the workgroups (a
                                     // 256-key chunk is 196 KB through ONE CU: the kernel is bound by that CU's share of the bandwidth);
                                     // 64 measured better per token on average and worse at 512 keys, where the merge in the
                                     // projection's prologue outgrows its first round of loads

// x = token_embedding + position_embedding (reference gpt.py:209-226 / 238-258): the token embedding is
// tok_emb[idx] or an explicit vector (`embeddings=`, emb [B, C]); the position embedding is pos_emb[pos],
// plus the vtokens_pos term (extra [B, C], gathered from vtokens_pos_emb by the caller) when given -- summed
// in the reference's order: tok + (pos + extra).
__global__ __launch_bounds__(256) void lm_embed_kernel(const int64_t *__restrict__ idx, const int32_t *__restrict__ pos,
                                                       const float *__restrict__ tok, const float *__restrict__ pe,
                                                       const float *__restrict__ emb, const float *__restrict__ extra,
                                                       float *__restrict__ x, int C, int vocab, int block_size) {
    const int b = blockIdx.x;
    int64_t id = idx ? idx[b] : 0;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int p = pos[b];
    p = p < 0 ? 0 : (p >= block_size ? block_size - 1 : p);
    const float *src = emb ? emb + (int64_t)b * C : tok + id * C;
    for (int i = threadIdx.x * 4; i < C; i += 256 * 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src + i);
        f32x4 c = *reinterpret_cast<const f32x4 *>(pe + (int64_t)p * C + i);
        if (extra) c = c + *reinterpret_cast<const f32x4 *>(extra + (int64_t)b * C + i);
        *reinterpret_cast<f32x4 *>(x + (int64_t)b * C + i) = a + c;
    }
}

// One workgroup = 4 waves x ROWS weight rows.  The activations (BQ rows of K floats, a few KB, L2-
// resident) are staged once per workgroup in LDS in panels of LM_KP floats -- layer-normalised on the
// way in when LN -- so the waves only stream weights from HBM: U chunks (256 floats) of each of the
// wave's ROWS rows are in flight while the previous U are multiplied (register double buffer).
//
// attention-merge staging (XM): x is not a tensor but the flash-decode chunk partials of
// lm_attn_decode_kernel; the workgroup merges them while it stages its activations, so the chunked
// attention needs neither a merge launch nor cross-workgroup fences.
int g_lm_wide_u = 2;  // "lm_wide_u": chunks per register buffer of the wide GEMVs (4 measured slower: 1.134 vs 1.114 ms / token)
constexpr int LM_KP = 2048;        // K panel staged in LDS (BQ * LM_KP * 4 B <= 64 KiB; one 6144-float panel for the
                                   // FC2 input at B = 1 measured the same: 1.114 ms / token)
constexpr int LM_MAX_CHUNKS = 32;  // attention chunks per sequence: max_len <= 8192
constexpr int LM_MAX_HEADS = 32;   // (bounds the merge-weight table in LDS: B * heads * chunks floats)

struct LmMerge {
    const float *part;         // [B][n_head][nchunk][2 + hd]
    const int32_t *cache_len;  // [B]
    int n_head, hd, nchunk;
    int chunk;                 // keys per attention chunk (LM_CHUNK, or LM_CHUNK_SHORT for caches of up to 4096 tokens)
    float *merged = nullptr;   // [B][n_head * hd] scratch: where the multi-stream MFMA path materialises the merged attention output
};

template <int BQ, int ROWS, int U, int ACT, bool LN, bool XM, int NW>
__global__ __launch_bounds__(NW * 64) void lm_gemv_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, const float *residual,
                                                      const float *__restrict__ g, const float *__restrict__ beta,
                                                      float *y, int N, int K, LmMerge mg) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [BQ][kp]
    __shared__ float s_mean[BQ], s_rstd[BQ];
    constexpr int NT = NW * 64;  // NW waves per workgroup: the launcher picks NW and ROWS so that the grid is a whole number of
                                 // workgroups per CU (a CU that gets one workgroup more than its neighbour streams that much longer)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = (blockIdx.x * NW + wave) * ROWS;
    const int kp = K < LM_KP ? K : LM_KP;
    const int nch = K >> 8;  // 256-float chunks
    const float *wr[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wr[r] = w + (int64_t)(row0 + r < N ? row0 + r : N - 1) * K + lane * 4;
    auto load_w = [&](f32x4 (&dst)[ROWS][U], int c0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + u < nch) {  // uniform; chunks past the row are neither loaded nor consumed
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    dst[r][u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(wr[r] + (c0 + u) * 256));
            }
        }
    };
    f32x4 wa[ROWS][U], wb[ROWS][U];
    load_w(wa, 0);  // in flight while the LayerNorm statistics / merge weights are computed
    // LayerNorm with the whole row in one LDS panel (K <= LM_KP, every GPT block): x, gamma and beta are requested
    // together at kernel start (one memory round trip instead of three dependent ones), the two-pass statistics are
    // block-wide reductions over registers, and the normalised row goes straight to LDS.
    const bool ln_fast = LN && K <= LM_KP;
    if (ln_fast) {
        __shared__ float s_part[2][BQ][NW];
        const int k4n = K >> 2;  // float4 per row (<= 512: NJ per thread)
        constexpr int NJ = (LM_KP / 4 + NT - 1) / NT;
        f32x4 xv[BQ][NJ], g4[NJ], b4[NJ];
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool has = tid + NT * j < k4n;
            g4[j] = has ? *reinterpret_cast<const f32x4 *>(g + (tid + NT * j) * 4) : zero4;
            b4[j] = has ? *reinterpret_cast<const f32x4 *>(beta + (tid + NT * j) * 4) : zero4;
#pragma unroll
            for (int b = 0; b < BQ; ++b)
                xv[b][j] = has ? *reinterpret_cast<const f32x4 *>(x + (int64_t)b * K + (tid + NT * j) * 4) : zero4;
        }
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            f32x4 t = xv[b][0];
#pragma unroll
            for (int j = 1; j < NJ; ++j) t = t + xv[b][j];
            const float sum = wave_allsum((t[0] + t[1]) + (t[2] + t[3]));
            if (lane == 0) s_part[0][b][wave] = sum;
        }
        lds_barrier();  // LDS traffic only: __syncthreads() would also wait for every weight load in flight
        float mean[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            float tot = 0.0f;
            if constexpr (NW == 4) tot = (s_part[0][b][0] + s_part[0][b][1]) + (s_part[0][b][2] + s_part[0][b][3]);
            else
#pragma unroll
                for (int w8 = 0; w8 < NW; ++w8) tot += s_part[0][b][w8];
            mean[b] = tot / (float)K;
            float q = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (tid + NT * j < k4n) {
                    const f32x4 a = xv[b][j] - mean[b];
                    q += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
                }
            }
            q = wave_allsum(q);
            if (lane == 0) s_part[1][b][wave] = q;
        }
        lds_barrier();  // LDS traffic only: __syncthreads() would also wait for every weight load in flight
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            float tot = 0.0f;
            if constexpr (NW == 4) tot = (s_part[1][b][0] + s_part[1][b][1]) + (s_part[1][b][2] + s_part[1][b][3]);
            else
#pragma unroll
                for (int w8 = 0; w8 < NW; ++w8) tot += s_part[1][b][w8];
            const float var = tot / (float)K;
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (tid + NT * j < k4n) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (xv[b][j][e] - mean[b]) * rstd * g4[j][e] + b4[j][e];
                    *reinterpret_cast<f32x4 *>(xs + b * kp + (tid + NT * j) * 4) = v;
                }
            }
        }
    } else if (LN) {
        // nn.LayerNorm statistics (two-pass, like ATen): wave w owns activation rows w and w + 4
        for (int b = wave; b < BQ; b += NW) {
            float sum = 0.0f;
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (int64_t)b * K + k);
                sum += (v[0] + v[1]) + (v[2] + v[3]);
            }
            const float mean = wave_allsum(sum) / (float)K;
            float q = 0.0f;
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (int64_t)b * K + k);
                const float a0 = v[0] - mean, a1 = v[1] - mean, a2 = v[2] - mean, a3 = v[3] - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
            const float var = wave_allsum(q) / (float)K;
            if (lane == 0) {
                s_mean[b] = mean;
                s_rstd[b] = 1.0f / sqrtf(var + 1e-5f);
            }
        }
    }
    float acc[ROWS][BQ];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < BQ; ++b) acc[r][b] = 0.0f;

    auto consume = [&](const f32x4 (&wv)[ROWS][U], int c0, int cbase) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + u < nch) {
                const int off = (c0 + u - cbase) * 256 + lane * 4;
#pragma unroll
                for (int b = 0; b < BQ; ++b) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + b * kp + off);
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) {
                        acc[r][b] = fmaf(wv[r][u][0], xv[0], acc[r][b]);
                        acc[r][b] = fmaf(wv[r][u][1], xv[1], acc[r][b]);
                        acc[r][b] = fmaf(wv[r][u][2], xv[2], acc[r][b]);
                        acc[r][b] = fmaf(wv[r][u][3], xv[3], acc[r][b]);
                    }
                }
            }
        }
    };

    constexpr int CPP = LM_KP / 256;  // chunks per panel (a multiple of 2 * U)
    for (int p0 = 0; p0 < nch; p0 += CPP) {
        lds_barrier();  // LDS traffic only: __syncthreads() would also wait for every weight load in flight  // previous panel fully consumed (and s_mean / s_rstd / s_f visible)
        const int pk = (nch - p0 < CPP ? nch - p0 : CPP) * 256;  // floats in this panel
        for (int i = tid * 4; i < BQ * pk && !ln_fast; i += NT * 4) {
            const int b = i / pk, k = i - b * pk;
            f32x4 v;
            if (XM) {
                // every staging thread merges its own head: global max over the chunks that hold keys, then
                // f_c = e^(M_c - M) / sum_c L_c e^(M_c - M).  Chunk 0 always holds keys, so its partial is requested
                // together with cache_len: up to 256 cached tokens the whole merge is one memory round trip (the
                // per-head weight table in LDS, its three dependent passes and its barrier are gone).
                const int col = p0 * 256 + k, h = col / mg.hd, d = col - h * mg.hd;  // hd % 4 == 0: one head per float4
                const int S = 2 + mg.hd;
                const float *pp = mg.part + ((int64_t)b * mg.n_head + h) * mg.nchunk * S;
                const float M0 = pp[0], L0 = pp[1];
                const float q00 = pp[2 + d], q01 = pp[3 + d], q02 = pp[4 + d], q03 = pp[5 + d];
                int used = (mg.cache_len[b] + 1 + mg.chunk - 1) / mg.chunk;
                if (used > mg.nchunk) used = mg.nchunk;  // a stream stepped past max_len (flagged by the attention kernel,
                                                         // which clamps the same way): never read past this head's partials
                float M = M0;
                for (int c = 1; c < used; ++c) M = fmaxf(M, pp[c * S]);
                float L = 0.0f;
                L += L0 * expf(M0 - M);
                for (int c = 1; c < used; ++c) L += pp[c * S + 1] * expf(pp[c * S] - M);
                const float f0 = expf(M0 - M) / L;
                v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                v[0] += q00 * f0;
                v[1] += q01 * f0;
                v[2] += q02 * f0;
                v[3] += q03 * f0;
                for (int c = 1; c < used; ++c) {
                    const float f = expf(pp[c * S] - M) / L;
                    const float *q = pp + c * S + 2 + d;
                    v[0] += q[0] * f;
                    v[1] += q[1] * f;
                    v[2] += q[2] * f;
                    v[3] += q[3] * f;
                }
            } else {
                v = *reinterpret_cast<const f32x4 *>(x + (int64_t)b * K + p0 * 256 + k);
            }
            if (LN) {
                const f32x4 g4 = *reinterpret_cast<const f32x4 *>(g + p0 * 256 + k);
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(beta + p0 * 256 + k);
                const float mean = s_mean[b], rstd = s_rstd[b];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * g4[e] + b4[e];
            }
            *reinterpret_cast<f32x4 *>(xs + b * kp + k) = v;
        }
        lds_barrier();  // LDS traffic only: __syncthreads() would also wait for every weight load in flight
        const int pend = p0 + CPP < nch ? p0 + CPP : nch;
        for (int c0 = p0; c0 < pend; c0 += 2 * U) {
            load_w(wb, c0 + U);  // may belong to the next panel: only the weights are prefetched
            consume(wa, c0, p0);
            if (c0 + U < pend) {
                load_w(wa, c0 + 2 * U);
                consume(wb, c0 + U, p0);
            } else {
                // wb holds the first group of the next panel: rotate it into wa
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int u = 0; u < U; ++u) wa[r][u] = wb[r][u];
            }
        }
    }
    if (row0 >= N) return;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < BQ; ++b) acc[r][b] = wave_allsum(acc[r][b]);
    // lane (r * BQ + b) finishes and stores output (b, row0 + r)
    float mine = 0.0f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < BQ; ++b)
            if (lane == r * BQ + b) mine = acc[r][b];
    if (lane < ROWS * BQ) {
        const int r = lane / BQ, b = lane % BQ, n = row0 + r;
        if (n < N) {
            float v = mine;
            if (bias) v += bias[n];
            if (ACT == 1) v = gelu_erf(v);
            if (residual) v += residual[(int64_t)b * N + n];
            y[(int64_t)b * N + n] = v;
        }
    }
}

// K-sliced GEMV (r05; B <= 2, K = NW * CW * 256): wave w of a workgroup owns columns [w * CW * 256, (w + 1) * CW * 256) of EVERY row the
// workgroup computes, so its slice of the activations lives in registers (no LDS staging, no panel barriers), a row's 6 (or 8) KiB
// are one contiguous request of the workgroup, and the grid is exactly one workgroup per CU with N / grid rows each.
// Per row: lane dot over its CW float4 -> wave sum -> one LDS word per (row, wave); one barrier at the end, then the first rows x BQ
// threads add the NW wave sums in wave order and apply bias / GELU / residual.
// What the timeline of a decode step (tools/r05/lm_timeline.py, profiles/r05_lm_timeline.txt) taught this kernel:
//   * loads return in order: what the prologue needs (activations, LayerNorm weights, the attention partials) is requested BEFORE the
//     weights, or it arrives behind 96 KiB of them;
//   * no load sits inside a branch -- not a uniform one either: at the join the compiler's counter model only knows "something may
//     be pending" and waits for vmcnt(0), i.e. for every weight in flight.  Short row groups clamp their row index instead (the
//     duplicate requests are L2 hits), bias / residual are loaded by every thread from a clamped index;
//   * the barriers order LDS only (lds_barrier): __syncthreads() drains vmcnt as well.
// Sums N = 2^k values over the 64 lanes of a wave in N - 1 + (6 - k) exchanges instead of 6 N: every halving step keeps half of the
// values (which half: one bit of the lane index, from bit 5 down) and adds the partner lane's copy of the kept half.  On return a[0] of
// lane l is the wave total of value (l >> (6 - k)); lanes with equal (l >> (6 - k)) hold the same total.
// partner lane's value for exchange distance S: DPP inside a row of 16 (S <= 8; S = 4 pairs lane j with j ^ 7, which differs in bit 2 --
// all a halving step needs), the LDS crossbar across rows
template <int S>
__device__ __forceinline__ float wave_xchg(float v) {
    if constexpr (S == 1) return dpp_f32<0xB1>(v);        // quad_perm [1, 0, 3, 2]
    else if constexpr (S == 2) return dpp_f32<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    else if constexpr (S == 4) return dpp_f32<0x141>(v);  // row_half_mirror
    else if constexpr (S == 8) return dpp_f32<0x128>(v);  // row_ror:8
    else return __shfl_xor(v, S);
}
template <int N, int S>
__device__ __forceinline__ void wave_transpose_step(float (&a)[N], int lane, int n) {  // n live values -> n / 2
    const bool up = (lane & S) != 0;
    const int half = n >> 1;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        if (i < half) {
            const float keep = up ? a[i + half] : a[i], send = up ? a[i] : a[i + half];
            a[i] = keep + wave_xchg<S>(send);
        }
    }
}
template <int N>
__device__ __forceinline__ void wave_transpose_sum(float (&a)[N], int lane) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "power of two");
    if constexpr (N >= 2) wave_transpose_step<N, 32>(a, lane, N);
    if constexpr (N >= 4) wave_transpose_step<N, 16>(a, lane, N / 2);
    if constexpr (N >= 8) wave_transpose_step<N, 8>(a, lane, N / 4);
    if constexpr (N >= 16) wave_transpose_step<N, 4>(a, lane, N / 8);
    if constexpr (N >= 32) wave_transpose_step<N, 2>(a, lane, N / 16);
    if constexpr (N >= 64) wave_transpose_step<N, 1>(a, lane, N / 32);
    // the 64 / N lanes that now hold partial sums of the same value: plain all-reduce (any pairing)
    if constexpr (N <= 32) a[0] += wave_xchg<1>(a[0]);
    if constexpr (N <= 16) a[0] += wave_xchg<2>(a[0]);
    if constexpr (N <= 8) a[0] += wave_xchg<4>(a[0]);
    if constexpr (N <= 4) a[0] += wave_xchg<8>(a[0]);
    if constexpr (N <= 2) a[0] += wave_xchg<16>(a[0]);
    if constexpr (N <= 1) a[0] += wave_xchg<32>(a[0]);
}

#ifndef LM_TRACE
#define LM_TRACE 0  // 1: s_memrealtime stamps per workgroup (tools/r05/lm_timeline.py; build with OMNITOK_EXTRA_FLAGS=-DLM_TRACE=1)
#endif
template <int BQ, int CW, int RG, int ACT, bool LN, bool XM, int NW>
__global__ __launch_bounds__(NW * 64) void lm_gemv_ks_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, const float *residual,
                                                             const float *__restrict__ g, const float *__restrict__ beta,
                                                             float *y, int N, int K, LmMerge mg, long long *trace) {
    extern __shared__ float ks_part[];  // [rows of this workgroup][BQ][NW]
    auto stamp = [&](int k) {
        if constexpr (LM_TRACE)
            if (trace && threadIdx.x == 0) trace[(int64_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r_begin = (int)((int64_t)blockIdx.x * N / gridDim.x), r_end = (int)((int64_t)(blockIdx.x + 1) * N / gridDim.x);
    const int nrows = r_end - r_begin, ngroups = (nrows + RG - 1) / RG;
    const int col0 = wave * CW * 256 + lane * 4;  // + cw * 256
    // the workgroup's rows through a buffer descriptor that ends with them: a request past the last row (short last group, the
    // prefetch of the groups after the last) returns zeros without touching memory -- no branch, no wasted traffic.  aux = 2: nt
    const auto w_rs = x3_rsrc(w + (int64_t)r_begin * K, nrows * K * 4);
    const int w_vo = col0 * 4;
    auto load_w = [&](f32x4 (&dst)[RG][CW], int grp) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int cw = 0; cw < CW; ++cw)
                // the row offset rides in the VECTOR offset: that operand (+ the instruction offset) is what the descriptor's range
                // check covers on every generation; an soffset is documented as outside it on some
                dst[r][cw] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_vo + ((grp * RG + r) * K + cw * 256) * 4, 0, 2));
    };
    f32x4 wa[RG][CW], wb[RG][CW];
    f32x4 xr[BQ][CW];
    // the finishing thread of output (row r, stream b) is thread r * BQ + b; every thread loads from a clamped index
    const int fr = tid / BQ < nrows ? tid / BQ : nrows - 1, fb = tid % BQ, fn = r_begin + fr;
    const float *bias_p = bias ? bias + fn : w, *res_p = residual ? residual + (int64_t)fb * N + fn : w;
    float e_bias, e_res;
    // ---- this wave's slice of the activations -> registers, then the weights of the first two row groups ----
    if constexpr (XM) {
        // merged attention partials (see lm_gemv_kernel): every lane merges the head its 4 columns belong to.  The partials of the
        // first XC chunks (1024 or 2048 keys) are requested before the weights, whether or not they hold keys (selected afterwards: their
        // addresses exist for every launch); only contexts beyond that take a second round of loads behind the weights.
        constexpr int XC = BQ <= 2 ? 8 : 2;  // (registers: 6 per chunk, stream and column group)
        const int S = 2 + mg.hd;
        const float *p0[BQ][CW];  // this head's partials: [M, L, o[hd]] per chunk
        int dd[CW];
        float Mc[BQ][CW][XC], Lc[BQ][CW][XC];
        f32x4 qc[BQ][CW][XC];
        int used[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            used[b] = mg.cache_len[b];
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) {
                const int col = col0 + cw * 256, h = col / mg.hd;
                dd[cw] = col - h * mg.hd;
                p0[b][cw] = mg.part + ((int64_t)b * mg.n_head + h) * mg.nchunk * S;
#pragma unroll
                for (int c = 0; c < XC; ++c) {
                    const float *pc = p0[b][cw] + (c < mg.nchunk ? c : mg.nchunk - 1) * S;
                    Mc[b][cw][c] = pc[0];
                    Lc[b][cw][c] = pc[1];
                    qc[b][cw][c] = f32x4{pc[2 + dd[cw]], pc[3 + dd[cw]], pc[4 + dd[cw]], pc[5 + dd[cw]]};
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // the requests above go out first (the scheduler otherwise hoists the address-ready weight loads)
        load_w(wa, 0);
        load_w(wb, 1);
        e_bias = *bias_p;
        e_res = *res_p;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            int nu = (used[b] + 1 + mg.chunk - 1) / mg.chunk;
            if (nu > mg.nchunk) nu = mg.nchunk;
#pragma unroll
            for (int cw = 0; cw < CW; ++cw) {
                const float *pb = p0[b][cw];
                float M = Mc[b][cw][0];
#pragma unroll
                for (int c = 1; c < XC; ++c) M = c < nu ? fmaxf(M, Mc[b][cw][c]) : M;
                for (int c = XC; c < nu; ++c) M = fmaxf(M, pb[c * S]);
                float L = 0.0f;
                L += Lc[b][cw][0] * expf(Mc[b][cw][0] - M);
#pragma unroll
                for (int c = 1; c < XC; ++c) L += c < nu ? Lc[b][cw][c] * expf(Mc[b][cw][c] - M) : 0.0f;
                for (int c = XC; c < nu; ++c) L += pb[c * S + 1] * expf(pb[c * S] - M);
                const float f0 = expf(Mc[b][cw][0] - M) / L;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                v[0] += qc[b][cw][0][0] * f0;
                v[1] += qc[b][cw][0][1] * f0;
                v[2] += qc[b][cw][0][2] * f0;
                v[3] += qc[b][cw][0][3] * f0;
#pragma unroll
                for (int c = 1; c < XC; ++c) {
                    if (c < nu) {  // (no loads inside: a select of registers)
                        const float f = expf(Mc[b][cw][c] - M) / L;
                        v[0] += qc[b][cw][c][0] * f;
                        v[1] += qc[b][cw][c][1] * f;
                        v[2] += qc[b][cw][c][2] * f;
                        v[3] += qc[b][cw][c][3] * f;
                    }
                }
                for (int c = XC; c < nu; ++c) {
                    const float f = expf(pb[c * S] - M) / L;
                    const float *q = pb + c * S + 2 + dd[cw];
                    v[0] += q[0] * f;
                    v[1] += q[1] * f;
                    v[2] += q[2] * f;
                    v[3] += q[3] * f;
                }
                xr[b][cw] = v;
            }
        }
    } else {
        f32x4 g4[CW], b4[CW];
        constexpr bool WAVE_LN = LN && BQ * NW * CW <= 16;  // statistics per wave over the whole row (while it fits registers) | LDS partials
        f32x4 xf[WAVE_LN ? BQ : 1][WAVE_LN ? NW * CW : 1];
        if constexpr (WAVE_LN) {
#pragma unroll
            for (int b = 0; b < BQ; ++b)
#pragma unroll
                for (int c = 0; c < NW * CW; ++c) xf[b][c] = *reinterpret_cast<const f32x4 *>(x + (int64_t)b * K + c * 256 + lane * 4);
        }
#pragma unroll
        for (int cw = 0; cw < CW; ++cw) {
#pragma unroll
            for (int b = 0; b < BQ; ++b) xr[b][cw] = *reinterpret_cast<const f32x4 *>(x + (int64_t)b * K + col0 + cw * 256);
            if constexpr (LN) {
                g4[cw] = *reinterpret_cast<const f32x4 *>(g + col0 + cw * 256);
                b4[cw] = *reinterpret_cast<const f32x4 *>(beta + col0 + cw * 256);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // the requests above go out first (the scheduler otherwise hoists the address-ready weight loads)
        load_w(wa, 0);
        load_w(wb, 1);
        e_bias = *bias_p;
        e_res = *res_p;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WAVE_LN) {
            // nn.LayerNorm, two-pass like ATen.  Every wave reduces the WHOLE row itself (xf: NW * CW float4 per lane, L2 hits for all but
            // the first wave to ask) instead of exchanging wave partials through LDS (no workgroup barrier in the prologue)
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                f32x4 t = xf[b][0];
#pragma unroll
                for (int c = 1; c < NW * CW; ++c) t = t + xf[b][c];
                const float mean = wave_allsum((t[0] + t[1]) + (t[2] + t[3])) / (float)K;
                float q = 0.0f;
#pragma unroll
                for (int c = 0; c < NW * CW; ++c) {
                    const f32x4 a = xf[b][c] - mean;
                    q += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
                }
                const float rstd = 1.0f / sqrtf(wave_allsum(q) / (float)K + 1e-5f);
#pragma unroll
                for (int cw = 0; cw < CW; ++cw)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[b][cw][e] = (xr[b][cw][e] - mean) * rstd * g4[cw][e] + b4[cw][e];
            }
        } else if constexpr (LN) {
            // more streams than whole rows fit the registers of: wave sums of the slices -> LDS -> every thread adds the NW partials
            __shared__ float s_red[2][BQ][NW];
            float part[BQ];
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                f32x4 t = xr[b][0];
#pragma unroll
                for (int cw = 1; cw < CW; ++cw) t = t + xr[b][cw];
                part[b] = (t[0] + t[1]) + (t[2] + t[3]);
            }
            wave_transpose_sum<BQ>(part, lane);
            if ((lane & (64 / BQ - 1)) == 0) s_red[0][lane / (64 / BQ)][wave] = part[0];
            lds_barrier();
            float mean[BQ];
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                float tot = 0.0f;
#pragma unroll
                for (int w8 = 0; w8 < NW; ++w8) tot += s_red[0][b][w8];
                mean[b] = tot / (float)K;
                float q = 0.0f;
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) {
                    const f32x4 a = xr[b][cw] - mean[b];
                    q += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
                }
                part[b] = q;
            }
            wave_transpose_sum<BQ>(part, lane);
            if ((lane & (64 / BQ - 1)) == 0) s_red[1][lane / (64 / BQ)][wave] = part[0];
            lds_barrier();
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                float tot = 0.0f;
#pragma unroll
                for (int w8 = 0; w8 < NW; ++w8) tot += s_red[1][b][w8];
                const float rstd = 1.0f / sqrtf(tot / (float)K + 1e-5f);
#pragma unroll
                for (int cw = 0; cw < CW; ++cw)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[b][cw][e] = (xr[b][cw][e] - mean[b]) * rstd * g4[cw][e] + b4[cw][e];
            }
        }
    }
    stamp(1);
    // ---- rows, RG at a time, two groups of weights in flight ----
    auto consume = [&](const f32x4 (&wv)[RG][CW], int grp) {
        float acc[RG * BQ];
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int b = 0; b < BQ; ++b) {
                float a = 0.0f;
#pragma unroll
                for (int cw = 0; cw < CW; ++cw) {
                    a = fmaf(wv[r][cw][0], xr[b][cw][0], a);
                    a = fmaf(wv[r][cw][1], xr[b][cw][1], a);
                    a = fmaf(wv[r][cw][2], xr[b][cw][2], a);
                    a = fmaf(wv[r][cw][3], xr[b][cw][3], a);
                }
                acc[r * BQ + b] = a;
            }
        wave_transpose_sum<RG * BQ>(acc, lane);  // lane l: the wave's sum for (row, stream) number l / (64 / (RG BQ))
        constexpr int LPV = 64 / (RG * BQ);      // lanes per value
        const int id = lane / LPV, r = id / BQ;
        if ((lane & (LPV - 1)) == 0 && grp * RG + r < nrows) ks_part[((grp * RG) * BQ + id) * NW + wave] = acc[0];
    };
    for (int grp = 0; grp < ngroups; grp += 2) {
        consume(wa, grp);
        if (grp == 0) stamp(2);
        load_w(wa, grp + 2);
        consume(wb, grp + 1);  // past the last group: nothing is written
        load_w(wb, grp + 3);
    }
    stamp(3);
    lds_barrier();
    if (tid < nrows * BQ) {
        float v = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) v += ks_part[(fr * BQ + fb) * NW + w8];
        if (bias) v += e_bias;
        if (ACT == 1) v = gelu_erf(v);
        if (residual) v += e_res;
        y[(int64_t)fb * N + fn] = v;
    }
    if constexpr (LM_TRACE) {
        __builtin_amdgcn_s_waitcnt(0);
        stamp(4);
    }
}

// F4 = head_dim / 32 float4 per lane: 8 lanes cover one K/V row of head_dim floats.  NWV waves per workgroup share the chunk's
// LM_CHUNK keys: with 8 waves a wave owns 32 keys = ONE batch of loads, and nothing in a load address depends on cache_len (rows are
// clamped to the slab, the new token's row comes from qkv and is selected afterwards): q, cache_len and every K / V row of the chunk
// are one memory round trip where the 4-wave form took three (q + length, then two batches of 32 keys) -- the kernel is a latency
// chain at decode sizes, not a stream (profiles/r05_lm_balance.txt).
template <int F4, int NWV, int CHUNK>
__global__ __launch_bounds__(NWV * 64) void lm_attn_decode_kernel(const float *__restrict__ qkv, float *kc, float *vc,
                                                                  const int32_t *__restrict__ cache_len, int n_head,
                                                                  int max_len, int prefill_T,
                                                                  float *__restrict__ part, int nchunk,
                                                                  int *__restrict__ err_flag) {
    constexpr int HD = F4 * 32;
    constexpr int KPW = CHUNK / NWV;  // keys per wave
    static_assert(KPW % 32 == 0, "a wave takes its keys in batches of 32");
    // decode: one query row per stream (row == b), len = cache_len[b].  prefill (prefill_T > 0): row =
    // b * T + t is query position t of stream b; its keys 0..t-1 are already in the cache (scattered by
    // lm_kv_scatter_kernel), key t is the row's own K/V
    const int c = blockIdx.x, h = blockIdx.y, row = blockIdx.z;
    const int b = prefill_T > 0 ? row / prefill_T : row;
    const int C = n_head * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 3, ds = lane & 7;  // key slot 0..7, dim slot 0..7
    const float *qn = qkv + (int64_t)row * 3 * C + h * HD;
    const float *kn = qn + C, *vn = qn + 2 * C;
    float *krow = kc + ((int64_t)b * n_head + h) * (int64_t)max_len * HD;
    float *vrow = vc + ((int64_t)b * n_head + h) * (int64_t)max_len * HD;
    const int kbase = c * CHUNK + wave * KPW;
    // everything this lane will read, requested at once
    f32x4 q[F4], knv[F4], vnv[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) {
        q[i] = *reinterpret_cast<const f32x4 *>(qn + (ds + 8 * i) * 4);
        knv[i] = *reinterpret_cast<const f32x4 *>(kn + (ds + 8 * i) * 4);
        vnv[i] = *reinterpret_cast<const f32x4 *>(vn + (ds + 8 * i) * 4);
    }
    f32x4 kk[4][F4], vv[4][F4];
    auto request = [&](int it0) {  // rows past the slab are read from row 0 (a safe address); rows past the sequence hold whatever
#pragma unroll                     // the cache holds (never used: masked below)
        for (int j = 0; j < 4; ++j) {
            const int key = kbase + (it0 + j) * 8 + slot;
            const int64_t off = (int64_t)(key < max_len ? key : 0) * HD;
#pragma unroll
            for (int i = 0; i < F4; ++i) {
                kk[j][i] = *reinterpret_cast<const f32x4 *>(krow + off + (ds + 8 * i) * 4);
                vv[j][i] = *reinterpret_cast<const f32x4 *>(vrow + off + (ds + 8 * i) * 4);
            }
        }
    };
    request(0);
    int len = prefill_T > 0 ? row - b * prefill_T : cache_len[row];
    if (len >= max_len) {
        // a caller stepped past the cache it allocated: flag it (omnitok_lm_overflowed) and stay inside
        // this head's K/V slab instead of reading the neighbour's rows
        if (err_flag && threadIdx.x == 0) *err_flag = 1;
        len = max_len - 1;
    }
    const int total = len + 1;
    if (c * CHUNK >= total) return;   // chunk beyond the sequence (static launch grid)
    // append the new token's K/V (only the workgroup whose chunk holds index len)
    if (prefill_T == 0 && len >= c * CHUNK && len < (c + 1) * CHUNK && len < max_len) {
        for (int d = tid; d < HD; d += NWV * 64) {
            krow[(int64_t)len * HD + d] = kn[d];
            vrow[(int64_t)len * HD + d] = vn[d];
        }
    }
    const float scale = 1.0f / sqrtf((float)HD);
    float m = -INFINITY, l = 0.0f;
    f32x4 o[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) o[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // 8 keys per iteration, four iterations (32 keys) per batch of loads
    for (int it0 = 0; it0 < KPW / 8; it0 += 4) {
        if (kbase + it0 * 8 >= total) break;  // wave-uniform: no keys left in this chunk
        if (it0 > 0) request(it0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = kbase + (it0 + j) * 8 + slot;
            const bool valid = key < total;
            // the new token's row comes from qkv (its cache copy is being written by this kernel)
            const bool fresh = key == len;
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < F4; ++i) {
                const f32x4 kv = fresh ? knv[i] : kk[j][i];
                s = fmaf(q[i][0], kv[0], s);
                s = fmaf(q[i][1], kv[1], s);
                s = fmaf(q[i][2], kv[2], s);
                s = fmaf(q[i][3], kv[3], s);
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s = valid ? s * scale : -INFINITY;
            const float mn = fmaxf(m, s);
            if (valid && mn > -INFINITY) {   // a masked key changes nothing (and its V row may be uninitialised)
                const float corr = expf(m - mn), p = expf(s - mn);
                l = l * corr + p;
#pragma unroll
                for (int i = 0; i < F4; ++i) o[i] = o[i] * corr + (fresh ? vnv[i] : vv[j][i]) * p;
                m = mn;
            }
        }
    }
    // merge the 8 key slots of the wave (lanes with equal dim slot: xor 8, 16, 32)
    float ma = m;
    ma = fmaxf(ma, __shfl_xor(ma, 8));
    ma = fmaxf(ma, __shfl_xor(ma, 16));
    ma = fmaxf(ma, __shfl_xor(ma, 32));
    const float sc = ma > -INFINITY ? expf(m - ma) : 0.0f;
    l *= sc;
#pragma unroll
    for (int i = 0; i < F4; ++i) o[i] = o[i] * sc;
#pragma unroll
    for (int sh = 8; sh <= 32; sh <<= 1) {
        l += __shfl_xor(l, sh);
#pragma unroll
        for (int i = 0; i < F4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[i][e] += __shfl_xor(o[i][e], sh);
    }
    // merge the waves through LDS
    __shared__ float s_m[NWV], s_l[NWV];
    __shared__ __attribute__((aligned(16))) float s_o[NWV][HD];
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < F4; ++i) *reinterpret_cast<f32x4 *>(&s_o[wave][(lane + 8 * i) * 4]) = o[i];
        if (lane == 0) {
            s_m[wave] = ma;
            s_l[wave] = l;
        }
    }
    __syncthreads();
    if (tid < HD) {
        float M = s_m[0];
#pragma unroll
        for (int wv = 1; wv < NWV; ++wv) M = fmaxf(M, s_m[wv]);
        float L = 0.0f, O = 0.0f;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) {
            const float f = s_m[wv] > -INFINITY ? expf(s_m[wv] - M) : 0.0f;
            L += s_l[wv] * f;
            O += s_o[wv][tid] * f;
        }
        float *pp = part + (((int64_t)row * n_head + h) * nchunk + c) * (2 + HD);
        pp[2 + tid] = O;
        if (tid == 0) {
            pp[0] = M;
            pp[1] = L;
        }
    }
}

__global__ void lm_attn_merge_kernel(const float *__restrict__ part, const int32_t *__restrict__ cache_len, int n_head,
                                     int HD, int nchunk, int prefill_T, float *__restrict__ out, int chunk) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;  // b: query row
    if (d >= HD) return;
    const int total = (prefill_T > 0 ? b % prefill_T : cache_len[b]) + 1;
    const int used = (total + chunk - 1) / chunk;
    const float *pp = part + ((int64_t)b * n_head + h) * nchunk * (2 + HD);
    float M = -INFINITY;
    for (int c = 0; c < used; ++c) M = fmaxf(M, pp[c * (2 + HD)]);
    float L = 0.0f, O = 0.0f;
    for (int c = 0; c < used; ++c) {
        const float f = expf(pp[c * (2 + HD)] - M);
        L += pp[c * (2 + HD) + 1] * f;
        O += pp[c * (2 + HD) + 2 + d] * f;
    }
    out[((int64_t)b * n_head + h) * HD + d] = O / L;
}

// ---- batched prefill of a conditioning prefix (the same arithmetic as T decode steps, as GEMMs) ------
// positions t < Te take the explicit embeddings emb[b, t] (prepended, gpt.py:214-216), the rest tok_emb[idx[b, t - Te]]
__global__ __launch_bounds__(256) void lm_embed_seq_kernel(const int64_t *__restrict__ idx, const float *__restrict__ tok,
                                                           const float *__restrict__ pe, const float *__restrict__ emb,
                                                           int Te, const float *__restrict__ extra,
                                                           float *__restrict__ x, int T, int C, int vocab) {
    const int64_t row = blockIdx.x;  // b * T + t
    const int t = (int)(row % T);
    const int64_t b = row / T;
    const float *src;
    if (t < Te) {
        src = emb + (b * Te + t) * C;
    } else {
        int64_t id = idx[b * (T - Te) + (t - Te)];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = tok + id * C;
    }
    for (int i = threadIdx.x * 4; i < C; i += 256 * 4) {
        f32x4 c = *reinterpret_cast<const f32x4 *>(pe + (int64_t)t * C + i);
        if (extra) c = c + *reinterpret_cast<const f32x4 *>(extra + row * C + i);
        *reinterpret_cast<f32x4 *>(x + row * C + i) = *reinterpret_cast<const f32x4 *>(src + i) + c;
    }
}

// nn.LayerNorm over rows of any width (two-pass, eps 1e-5): one wave per row
__global__ __launch_bounds__(256) void lm_layernorm_rows_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                const float *__restrict__ beta, float *__restrict__ y,
                                                                int64_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * K;
    float s = 0.0f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + k);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_allsum(s) / (float)K;
    float q = 0.0f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + k);
        const float a0 = v[0] - mean, a1 = v[1] - mean, a2 = v[2] - mean, a3 = v[3] - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = 1.0f / sqrtf(wave_allsum(q) / (float)K + 1e-5f);
    for (int k = lane * 4; k < K; k += 256) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(xr + k);
        const f32x4 g4 = *reinterpret_cast<const f32x4 *>(g + k), b4 = *reinterpret_cast<const f32x4 *>(beta + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * g4[e] + b4[e];
        *reinterpret_cast<f32x4 *>(y + row * K + k) = v;
    }
}

__global__ __launch_bounds__(256) void lm_gelu_kernel(float *__restrict__ x, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<f32x4 *>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    reinterpret_cast<f32x4 *>(x)[i] = v;
}

// qkv[B*T, 3C] (query | key | value) -> K/V cache rows [b][h][t][hd]
__global__ __launch_bounds__(256) void lm_kv_scatter_kernel(const float *__restrict__ qkv, float *__restrict__ kc,
                                                            float *__restrict__ vc, int T, int n_head, int hd,
                                                            int max_len) {
    const int64_t row = blockIdx.x;  // b * T + t
    const int b = (int)(row / T), t = (int)(row % T), C = n_head * hd;
    for (int i = threadIdx.x * 4; i < C; i += 256 * 4) {
        const int h = i / hd, d = i - h * hd;
        const int64_t dst = (((int64_t)b * n_head + h) * max_len + t) * hd + d;
        *reinterpret_cast<f32x4 *>(kc + dst) = *reinterpret_cast<const f32x4 *>(qkv + row * 3 * C + C + i);
        *reinterpret_cast<f32x4 *>(vc + dst) = *reinterpret_cast<const f32x4 *>(qkv + row * 3 * C + 2 * C + i);
    }
}

__global__ void lm_set_len_kernel(int32_t *pos, int32_t *cache_len, int B, int T) {
    const int b = threadIdx.x;
    if (b < B) {
        pos[b] = T;
        cache_len[b] = T;
    }
}

__global__ void lm_advance_kernel(int32_t *pos, int32_t *cache_len, int B) {
    const int b = threadIdx.x;
    if (b < B) {
        pos[b] += 1;
        cache_len[b] += 1;
    }
}

// rows of a (key | query | value) trio concatenated as (query | key | value) [3C, C] / [3C]
__global__ void lm_concat3_kernel(const float *a, const float *b, const float *c, int64_t n, float *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    out[i] = i < n ? a[i] : (i < 2 * n ? b[i - n] : c[i - 2 * n]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-stream decode GEMV on the matrix cores (r06): y[b][n] = act(LN(x_b) . w_n + bias_n) (+ residual) for 4 .. 8 streams in ONE
// pass over the weights.  The K-sliced VALU kernel above spends, per weight float4, 4 FMAs per stream plus a cross-lane
// transpose-sum per row and stream: 8 streams cost 1.43 ms per decode step against 0.85 ms for one, although the weight bytes are
// the same.  Here v_mfma_f32_4x4x1_16B_f32 does the work: 16 independent 4 x 4 outer products per instruction, fp32 in, fp32
// accumulate -- block b of a lane group holds k = k0 + 4 b + e, so one instruction multiplies 4 weight rows x 16 k by 16 k x 4 streams.
//   * workgroup = 8 waves = a contiguous slice of the weight rows (grid = a multiple of the CUs: every CU streams the same bytes,
//     as in lm_gemv_ks_kernel), in quads of 4 rows; lane (blk = lane / 4, r = lane % 4) loads w[row 4 q + r][k0 + 64 c + 4 blk .. + 3]:
//     256 contiguous bytes per row and instruction.  Wave w owns k in [w K / 8, (w + 1) K / 8) = CH chunks of 64;
//   * ALL of the workgroup's weights (<= 24 loads of 16 B per lane = 24 KiB per wave) are requested in one burst at the start --
//     rows beyond the slice are cut off by the buffer descriptor -- and land while the LayerNorm statistics are computed.
//     (A first form on 16-row v_mfma_f32_16x16x4_f32 tiles kept one 1536-k pass of a 16-row group in flight per wave: with 6 live
//     rows of 16 per workgroup at N = 1536 it streamed 1.3 TB/s, 2.2 ms per step -- profiles/r06_lm_mfma.txt);
//   * activations: lane (blk, c) holds x[stream 4 sq + c][the same k] for sq = 0, 1; with a LayerNorm (K = 1536: 24 registers) they
//     are loaded BEFORE the weights, normalised in registers (two-pass statistics across the workgroup) and kept; without one they
//     are read chunk by chunk (L2) behind the weights;
//   * per row quad and stream quad one 4-register accumulator; at the end the 16 k blocks are summed across lanes (4 xor steps),
//     the 8 waves through LDS in wave order (deterministic), then bias / GELU / residual.
// Same fp32 products as the VALU kernels, another summation order (tests/test_gpu_lm.py::test_gemv_mfma_path_equals_valu_path).
constexpr int LMM_NW = 8;
int g_lm_mfma_mult = 1;  // "lm_mfma_mult": workgroups per CU of lm_gemm4_kernel (at least; more when a slice would exceed QMAX quads)
// "lm_mfma": 0 (default) = the VALU kernels | 1 = groups of 4 .. 8 streams take lm_gemm4_kernel when K / 512 is 3, 4, 12 or 16.
// Measured (profiles/r06_lm_mfma.txt): 8 streams 1.73 - 1.83 ms per decode step against 1.42 ms for the VALU kernels -- the one-burst
// form pays transfer + compute in series (13 - 14 us per GEMV whatever its bytes) where the K-sliced VALU kernel overlaps them, and
// the merged attention output costs a launch of its own; correct (tests) and kept as an A/B arm, not the default.
int g_lm_mfma = 0;

template <int ACT, bool LN, int CH, int QMAX>
__global__ __launch_bounds__(LMM_NW * 64, 2) void lm_gemm4_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                  const float *__restrict__ bias, const float *residual,
                                                                  const float *__restrict__ g, const float *__restrict__ beta,
                                                                  float *y, int B, int N, int K) {
    static_assert(CH * QMAX <= 24 && (!LN || CH <= 4), "registers: weights in one burst, LayerNorm activations resident");
    // LDS: [NW * 4 lane rows][QMAX * 4 weight rows][8 streams] partial sums | [NW * 4][8] LayerNorm partials
    extern __shared__ float mm_lds[];
    constexpr int NP = LMM_NW * 4;               // partial sums per output: 8 waves x the 4 sixteen-lane rows of a wave
    float *part = mm_lds, *red = mm_lds + NP * QMAX * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = lane >> 2, r = lane & 3, lrow = lane >> 4;
    const int r_begin = (int)((int64_t)blockIdx.x * N / gridDim.x), r_end = (int)((int64_t)(blockIdx.x + 1) * N / gridDim.x);
    const int nrows = r_end - r_begin, Q = (nrows + 3) >> 2;   // <= QMAX by the launcher's grid
    const int k0 = wave * (CH * 64) + blk * 4;                 // this lane's first k
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    // streams of this lane: r and r + 4 (those beyond B repeat the last one and are never stored)
    const float *x0 = x + (int64_t)(r < B ? r : B - 1) * K + k0, *x1 = x + (int64_t)(r + 4 < B ? r + 4 : B - 1) * K + k0;
    // every load of the kernel goes out here, activations and LayerNorm constants first (they are needed first)
    f32x4v xa[CH][2], gv[LN ? CH : 1], bv[LN ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        xa[c][0] = *reinterpret_cast<const f32x4v *>(x0 + c * 64);
        xa[c][1] = *reinterpret_cast<const f32x4v *>(x1 + c * 64);
        if constexpr (LN) {
            gv[c] = *reinterpret_cast<const f32x4v *>(g + k0 + c * 64);
            bv[c] = *reinterpret_cast<const f32x4v *>(beta + k0 + c * 64);
        }
    }
    // weights: rows beyond the workgroup's slice (and whole quads beyond Q) lie outside the descriptor -- zeros, no traffic, no branch
    const auto w_rs = x3_rsrc(w + (int64_t)r_begin * K, nrows * K * 4);
    f32x4v wv[QMAX][CH];
#pragma unroll
    for (int q = 0; q < QMAX; ++q)
#pragma unroll
        for (int c = 0; c < CH; ++c)
            wv[q][c] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(w_rs, ((q * 4 + r) * K + k0 + c * 64) * 4, 0, 2));
    __builtin_amdgcn_sched_barrier(0);   // the burst stays a burst: the scheduler otherwise sinks each load to its first use
    if constexpr (LN) {
        // two-pass statistics of streams r, r + 4 over all K: the 4 blocks of a 16-lane row by DPP, the 32 rows of the workgroup
        // through LDS.  lds_barrier (lgkmcnt only): __syncthreads() would drain vmcnt -- the weight burst -- in front of the first one
        float s[2] = {0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int sq = 0; sq < 2; ++sq) s[sq] += (xa[c][sq][0] + xa[c][sq][1]) + (xa[c][sq][2] + xa[c][sq][3]);
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
            s[sq] += dpp_f32<0x128>(s[sq]);   // row_ror:8, row_ror:4
            s[sq] += dpp_f32<0x124>(s[sq]);
            if ((lane & 12) == 0) red[(wave * 4 + lrow) * 8 + sq * 4 + r] = s[sq];
        }
        lds_barrier();
        float mean[2], rstd[2];
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
            float t = 0.0f;
#pragma unroll
            for (int i = 0; i < NP; ++i) t += red[i * 8 + sq * 4 + r];
            mean[sq] = t / (float)K;
        }
        lds_barrier();
        float qs[2] = {0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int sq = 0; sq < 2; ++sq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = xa[c][sq][e] - mean[sq];
                    qs[sq] = fmaf(d, d, qs[sq]);
                }
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
            qs[sq] += dpp_f32<0x128>(qs[sq]);
            qs[sq] += dpp_f32<0x124>(qs[sq]);
            if ((lane & 12) == 0) red[(wave * 4 + lrow) * 8 + sq * 4 + r] = qs[sq];
        }
        lds_barrier();
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) {
            float t = 0.0f;
#pragma unroll
            for (int i = 0; i < NP; ++i) t += red[i * 8 + sq * 4 + r];
            rstd[sq] = 1.0f / sqrtf(t / (float)K + 1e-5f);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int sq = 0; sq < 2; ++sq)
#pragma unroll
                for (int e = 0; e < 4; ++e) xa[c][sq][e] = (xa[c][sq][e] - mean[sq]) * rstd[sq] * gv[c][e] + bv[c][e];
    }
    f32x4v acc[QMAX][2];
#pragma unroll
    for (int q = 0; q < QMAX; ++q)
#pragma unroll
        for (int sq = 0; sq < 2; ++sq) acc[q][sq] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    // (e before q: consecutive MFMAs go to the 2 QMAX different accumulators, none waits for its predecessor's result;
    //  quads beyond Q multiply zeros -- cheaper than a branch per pair of MFMAs)
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
#pragma unroll
                for (int sq = 0; sq < 2; ++sq)
                    acc[q][sq] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[q][c][e], xa[c][sq][e], acc[q][sq], 0, 0, 0);
    // acc[q][sq][v] (lane = (blk, c)): row 4 q + v, stream 4 sq + c, partial over the k = 4 blk + e (mod 64) of this wave's slice.
    // The 4 blocks of each 16-lane row are summed by DPP; the 4 rows of the wave and the 8 waves meet in LDS.
#pragma unroll
    for (int q = 0; q < QMAX; ++q)
#pragma unroll
        for (int sq = 0; sq < 2; ++sq)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float t = acc[q][sq][v];
                t += dpp_f32<0x128>(t);
                t += dpp_f32<0x124>(t);
                if ((lane & 12) == 0) part[((wave * 4 + lrow) * QMAX * 4 + q * 4 + v) * 8 + sq * 4 + r] = t;
            }
    lds_barrier();
    for (int o = tid; o < Q * 32; o += LMM_NW * 64) {
        const int row = o >> 3, st = o & 7, n = r_begin + row;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; ++i) v += part[i * QMAX * 32 + o];
        if (st < B && n < r_end) {
            if (bias) v += bias[n];
            if (ACT == 1) v = gelu_erf(v);
            if (residual) v += residual[(int64_t)st * N + n];
            y[(int64_t)st * N + n] = v;
        }
    }
}

template <int ACT, bool LN, int CH, int QMAX>
static void launch_gemm4_cfg(const float *x, const float *w, const float *bias, const float *residual, const float *g, const float *beta,
                             float *y, int B, int N, int K, hipStream_t stream) {
    int n_cu = 256;
    (void)current_device_cus(&n_cu);
    const int per = QMAX * 4;                               // rows a workgroup can own
    int mult = g_lm_mfma_mult > 1 ? g_lm_mfma_mult : 1;
    while ((int64_t)n_cu * mult * per < (int64_t)N + n_cu * mult) ++mult;   // ceil(N / grid) <= per
    int grid = n_cu * mult;
    if (grid > N) grid = N;
    hipLaunchKernelGGL((lm_gemm4_kernel<ACT, LN, CH, QMAX>), dim3(grid), dim3(LMM_NW * 64), (LMM_NW * 4 * QMAX * 32 + LMM_NW * 4 * 8) * 4, stream,
                       x, w, bias, residual, g, beta, y, B, N, K);
}

// K / 512 chunks of 64 k per wave: 3 (K = 1536), 4 (2048), 12 (6144), 16 (8192); false: no MFMA form for this call
template <int ACT, bool LN>
static bool launch_gemm4(const float *x, const float *w, const float *bias, const float *residual, const float *g, const float *beta,
                         float *y, int B, int N, int K, hipStream_t stream) {
    switch (K) {
        case 1536: launch_gemm4_cfg<ACT, LN, 3, 8>(x, w, bias, residual, g, beta, y, B, N, K, stream); return true;
        case 2048: launch_gemm4_cfg<ACT, LN, 4, 6>(x, w, bias, residual, g, beta, y, B, N, K, stream); return true;
        default: break;
    }
    if constexpr (!LN) {   // (the wide inputs -- FC2's 4 C -- carry no LayerNorm)
        switch (K) {
            case 6144: launch_gemm4_cfg<ACT, false, 12, 2>(x, w, bias, residual, g, beta, y, B, N, K, stream); return true;
            case 8192: launch_gemm4_cfg<ACT, false, 16, 1>(x, w, bias, residual, g, beta, y, B, N, K, stream); return true;
            default: break;
        }
    }
    return false;
}

int g_lm_attn_waves = 8;  // "lm_attn_waves": 8 = a wave owns 32 keys of the 256-key chunk (one batch of loads) | 4 = 64 keys, two batches
int g_lm_attn_short = 1;  // "lm_attn_short": 1 = 128-key attention chunks for caches of up to 4096 tokens (read at omnitok_lm_alloc_cache) | 0 = 256
int g_lm_ks_deep = 0;   // "lm_ks_deep": weights in flight per wave of the K-sliced GEMV: 1 = 16 KiB | 0 = 8 KiB
int g_lm_ksliced = 2;  // "lm_ksliced": 2 = lm_gemv_ks_kernel for K in {1536, 2048, 6144, 8192} | 1 = only for B <= 2 | 0 = lm_gemv_kernel
int g_lm_balance = 1;  // "lm_balance": 1 = waves per workgroup and rows per wave chosen so that every CU gets the same number of
                       // workgroups | 0 = 4 waves, 1 row (N <= 2048) or 2 rows per wave

template <int BQ, int ROWS, int U, int ACT, bool LN, bool XM, int NW>
static void launch_gemv_cfg(const float *x, const float *w, const float *bias, const float *residual, const float *g,
                            const float *beta, float *y, int N, int K, const LmMerge &mg, hipStream_t stream) {
    const int kp = K < LM_KP ? K : LM_KP;
    const int lds = BQ * kp * 4;
    const int rows_per_wg = NW * ROWS;
    if (lds > 65536) (void)set_max_dynamic_lds(reinterpret_cast<const void *>(lm_gemv_kernel<BQ, ROWS, U, ACT, LN, XM, NW>), lds);
    hipLaunchKernelGGL((lm_gemv_kernel<BQ, ROWS, U, ACT, LN, XM, NW>), dim3((N + rows_per_wg - 1) / rows_per_wg), dim3(NW * 64),
                       lds, stream, x, w, bias, residual, g, beta, y, N, K, mg);
}

extern long long *g_gemm_trace;
static int g_lm_trace_launch = 0;  // launches since the trace buffer was set (omnitok_debug_set_gemm_trace resets it through lm_trace_reset)
void lm_trace_reset() { g_lm_trace_launch = 0; }
static long long *lm_trace_slot(int grid) {  // 8 stamps per workgroup, 1024 workgroups per launch slot, 256 slots
    if (!g_gemm_trace || g_lm_trace_launch >= 256 || grid > 1024) return nullptr;
    return g_gemm_trace + (int64_t)(g_lm_trace_launch++) * 1024 * 8;
}

template <int BQ, int ACT, bool LN, bool XM>
static void launch_gemv_rows(const float *x, const float *w, const float *bias, const float *residual, const float *g,
                             const float *beta, float *y, int N, int K, const LmMerge &mg, hipStream_t stream) {
    // ROWS weight rows x U chunks (x2 register buffers) per wave = 8 KiB of weights in flight per wave.
    // Narrow outputs (N <= 2048: the C x C and C x 4C projections) take one row per wave so that the
    // launch still covers every CU.  Wide outputs: 2 rows x 2 chunks x 2 buffers per wave, i.e. a whole K = 1536 row pair is
    // requested before the LayerNorm prologue finishes (the matrix streams from HBM while x is normalised).
    // Balance (r05): a decode GEMV lasts 5-10 us and every CU streams at its own ~25-30 GB/s, so the launch takes as long as the
    // CU with the most workgroups: 384 workgroups on 256 CUs (N = 1536 at 4 rows per workgroup) run like 512.  With 6 waves per
    // workgroup N = 1536 is exactly one workgroup per CU and N = 4608 exactly three (profiles/r05_lm_balance.txt).
    int n_cu = 256;
    (void)current_device_cus(&n_cu);
    if (BQ <= 2 ? g_lm_ksliced >= 1 : g_lm_ksliced >= 2) {
        // K-sliced kernel: one workgroup per CU (more only when a workgroup would own more than 64 rows or 384 outputs)
        int mult = 1;  // workgroups per CU: 2 and 3 measured 7 % and 15 % slower per token (every workgroup repeats the prologue)
        while (N / (n_cu * mult) > 64 / (BQ > 4 ? 2 : 1)) ++mult;
        while (mult > 1 && N < n_cu * mult) --mult;
        const int grid = n_cu * mult;
        const int max_rows = (N + grid - 1) / grid + 1;
        if (N >= grid && max_rows * BQ <= 6 * 64) {
            const int lds = max_rows * BQ * 8 * 4;
#define OT_KS(CW_, RG_, NW_)                                                                                                         \
    hipLaunchKernelGGL((lm_gemv_ks_kernel<BQ, CW_, RG_, ACT, LN, XM, NW_>), dim3(grid), dim3(NW_ * 64), lds, stream, x, w, bias, residual, g, \
                       beta, y, N, K, mg, lm_trace_slot(grid))
            // rows per group: 2 groups in flight per wave = 8 KiB ("lm_ks_deep" 1: 16 KiB, measured slower: profiles/r05_lm_timeline.txt)
            constexpr int DEEP = BQ <= 2 ? 2 : 1;  // (the deep variants exist for B <= 2 only)
            const bool deep = g_lm_ks_deep == 1 && BQ <= 2;
            if (K == 6 * 256) {
                if (deep) OT_KS(1, 4 * DEEP, 6);
                else OT_KS(1, 4, 6);
                return;
            }
            if constexpr (!XM) {  // (the attention merge only feeds the C x C projection)
                if (K == 6 * 4 * 256) {
                    if (deep) OT_KS(4, DEEP, 6);
                    else OT_KS(4, 1, 6);
                    return;
                }
                if (K == 8 * 4 * 256) {
                    if (deep) OT_KS(4, DEEP, 8);
                    else OT_KS(4, 1, 8);
                    return;
                }
            }
            if (K == 8 * 256) {
                if (deep) OT_KS(1, 4 * DEEP, 8);
                else OT_KS(1, 4, 8);
                return;
            }
#undef OT_KS
        }
    }
    auto waste = [&](int rows_per_wg) {  // workgroup slots of the last round that stay empty, as a fraction of the launch
        const int wgs = (N + rows_per_wg - 1) / rows_per_wg;
        const int rounds = (wgs + n_cu - 1) / n_cu;
        return (double)(rounds * n_cu - wgs) / (double)(rounds * n_cu);
    };
    const bool six = g_lm_balance && BQ <= 2 && K % 256 == 0;
    if (N <= 2048) {
        if (six && waste(6) + 0.05 < waste(4)) launch_gemv_cfg<BQ, 1, 4, ACT, LN, XM, 6>(x, w, bias, residual, g, beta, y, N, K, mg, stream);
        else launch_gemv_cfg<BQ, 1, 4, ACT, LN, XM, 4>(x, w, bias, residual, g, beta, y, N, K, mg, stream);
    } else if (six && waste(6) + 0.05 < waste(8)) {
        launch_gemv_cfg<BQ, 1, 4, ACT, LN, XM, 6>(x, w, bias, residual, g, beta, y, N, K, mg, stream);
    } else if (g_lm_wide_u == 2) {
        launch_gemv_cfg<BQ, 2, 2, ACT, LN, XM, 4>(x, w, bias, residual, g, beta, y, N, K, mg, stream);
    } else {
        launch_gemv_cfg<BQ, 2, 4, ACT, LN, XM, 4>(x, w, bias, residual, g, beta, y, N, K, mg, stream);
    }
}

template <int BQ>
static void launch_gemv_bq(const float *x, const float *w, const float *bias, const float *residual, const float *g,
                           const float *beta, float *y, int N, int K, int act, const LmMerge *mg, hipStream_t stream) {
    const LmMerge none{nullptr, nullptr, 0, 0, 0, LM_CHUNK};
    if (mg) {
        launch_gemv_rows<BQ, 0, false, true>(x, w, bias, residual, g, beta, y, N, K, *mg, stream);
    } else if (g) {
        if (act) launch_gemv_rows<BQ, 1, true, false>(x, w, bias, residual, g, beta, y, N, K, none, stream);
        else launch_gemv_rows<BQ, 0, true, false>(x, w, bias, residual, g, beta, y, N, K, none, stream);
    } else {
        if (act) launch_gemv_rows<BQ, 1, false, false>(x, w, bias, residual, g, beta, y, N, K, none, stream);
        else launch_gemv_rows<BQ, 0, false, false>(x, w, bias, residual, g, beta, y, N, K, none, stream);
    }
}

// y = act(LN(x) W^T + b) (+ residual) for B rows, in groups of up to 8 rows per pass over the weights.
// mg != nullptr: the activations are the merged attention partials (x unused).
static int lm_gemv_any(const float *x, const float *w, const float *bias, const float *residual, const float *g,
                       const float *beta, float *y, int B, int N, int K, int act, const LmMerge *mg, hipStream_t stream) {
    for (int b0 = 0; b0 < B;) {
        const int left = B - b0;
        const float *xb = x ? x + (int64_t)b0 * K : nullptr, *rb = residual ? residual + (int64_t)b0 * N : nullptr;
        float *yb = y + (int64_t)b0 * N;
        const bool mfma_k = K == 1536 || K == 2048 || (!g && (K == 6144 || K == 8192));
        if (g_lm_mfma && left >= 4 && mfma_k && N >= 16 && (int64_t)N * K < (1ll << 29) && (!mg || mg->merged)) {
            // 4 .. 8 streams in one pass over the weights on the matrix cores (lm_gemm4_kernel)
            const int bq = left > 8 ? 8 : left;
            if (mg) {   // the merged attention output as a tensor first (the VALU path merges inside its prologue)
                hipLaunchKernelGGL(lm_attn_merge_kernel, dim3(mg->n_head, bq), dim3(128), 0, stream,
                                   mg->part + (int64_t)b0 * mg->n_head * mg->nchunk * (2 + mg->hd), mg->cache_len + b0, mg->n_head, mg->hd,
                                   mg->nchunk, 0, mg->merged + (int64_t)b0 * K, mg->chunk);
                xb = mg->merged + (int64_t)b0 * K;
            }
            bool ok;
            if (g) ok = act ? launch_gemm4<1, true>(xb, w, bias, rb, g, beta, yb, bq, N, K, stream)
                            : launch_gemm4<0, true>(xb, w, bias, rb, g, beta, yb, bq, N, K, stream);
            else ok = act ? launch_gemm4<1, false>(xb, w, bias, rb, g, beta, yb, bq, N, K, stream)
                          : launch_gemm4<0, false>(xb, w, bias, rb, g, beta, yb, bq, N, K, stream);
            if (ok) {
                OT_LAUNCH_CHECK("lm_gemm4");
                b0 += bq;
                continue;
            }
        }
        const int bq = left >= 8 ? 8 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1));
        LmMerge m2{nullptr, nullptr, 0, 0, 0, LM_CHUNK};
        if (mg) {
            m2 = *mg;
            m2.part += (int64_t)b0 * mg->n_head * mg->nchunk * (2 + mg->hd);
            m2.cache_len += b0;
        }
        const LmMerge *mp = mg ? &m2 : nullptr;
        switch (bq) {
            case 8: launch_gemv_bq<8>(xb, w, bias, rb, g, beta, yb, N, K, act, mp, stream); break;
            case 4: launch_gemv_bq<4>(xb, w, bias, rb, g, beta, yb, N, K, act, mp, stream); break;
            case 2: launch_gemv_bq<2>(xb, w, bias, rb, g, beta, yb, N, K, act, mp, stream); break;
            default: launch_gemv_bq<1>(xb, w, bias, rb, g, beta, yb, N, K, act, mp, stream); break;
        }
        OT_LAUNCH_CHECK("lm_gemv");
        b0 += bq;
    }
    return OMNITOK_OK;
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_lm_gemv(const float *x, const float *w, const float *bias, const float *residual,
                               const float *ln_gamma, const float *ln_beta, float *y, int B, int N, int K, int act,
                               omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG(x && w && y, "lm_gemv: null pointer");
    OT_CHECK_ARG(B > 0 && B <= 16 && N > 0 && K > 0 && K % 256 == 0, "lm_gemv: bad sizes B=%d N=%d K=%d", B, N, K);
    OT_CHECK_ARG((ln_gamma != nullptr) == (ln_beta != nullptr), "lm_gemv: ln_gamma / ln_beta must come together");
    OT_CHECK_ARG(act == 0 || act == 1, "lm_gemv: act %d", act);
    OT_CHECK_ARG(aligned16(x) && aligned16(w), "lm_gemv: unaligned");
    return lm_gemv_any(x, w, bias, residual, ln_gamma, ln_beta, y, B, N, K, act, nullptr, stream);
}

// chunk partials only (the engine merges them inside the proj GEMV)
// prefill_T > 0: B * prefill_T query rows (row = b * T + t, keys 0..t), chunks sized for T keys
static int lm_attn_partials(const float *qkv, float *kc, float *vc, const int32_t *cache_len, int B, int n_head,
                            int head_dim, int max_len, float *scratch, hipStream_t stream, int prefill_T = 0,
                            int *err_flag = nullptr, int chunk = LM_CHUNK) {
    OT_CHECK_ARG(qkv && kc && vc && (cache_len || prefill_T > 0) && scratch, "lm_attn_decode: null pointer");
    OT_CHECK_ARG(head_dim == 64 || head_dim == 96 || head_dim == 128, "lm_attn_decode: head_dim %d (64 | 96 | 128)",
                 head_dim);
    OT_CHECK_ARG(max_len > 0 && n_head > 0, "lm_attn_decode: bad sizes");
    const int nchunk = ((prefill_T > 0 ? prefill_T : max_len) + chunk - 1) / chunk;
    const int rows = prefill_T > 0 ? B * prefill_T : B;
    OT_CHECK_ARG(rows <= 65535, "lm_attn_decode: %d query rows (max 65535)", rows);
    const dim3 grid(nchunk, n_head, rows);
    const bool wide = g_lm_attn_waves == 8;
#define OT_LM_ATTN(F4_)                                                                                                                    \
    if (chunk == LM_CHUNK_SHORT)                                                                                                          \
        hipLaunchKernelGGL((lm_attn_decode_kernel<F4_, 4, LM_CHUNK_SHORT>), grid, dim3(256), 0, stream, qkv, kc, vc, cache_len, n_head,  \
                           max_len, prefill_T, scratch, nchunk, err_flag);                                                               \
    else if (wide)                                                                                                                        \
        hipLaunchKernelGGL((lm_attn_decode_kernel<F4_, 8, LM_CHUNK>), grid, dim3(512), 0, stream, qkv, kc, vc, cache_len, n_head, max_len, \
                           prefill_T, scratch, nchunk, err_flag);                                                                        \
    else                                                                                                                                  \
        hipLaunchKernelGGL((lm_attn_decode_kernel<F4_, 4, LM_CHUNK>), grid, dim3(256), 0, stream, qkv, kc, vc, cache_len, n_head, max_len, \
                           prefill_T, scratch, nchunk, err_flag)
    switch (head_dim) {
        case 64: OT_LM_ATTN(2); break;
        case 96: OT_LM_ATTN(3); break;
        default: OT_LM_ATTN(4); break;
    }
#undef OT_LM_ATTN
    OT_LAUNCH_CHECK("lm_attn_decode");
    return OMNITOK_OK;
}

extern "C" int omnitok_lm_attn_decode(const float *qkv, float *kc, float *vc, const int32_t *cache_len, int B,
                                      int n_head, int head_dim, int max_len, float *scratch, float *out,
                                      omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG(out, "lm_attn_decode: null pointer");
    if (int rc = lm_attn_partials(qkv, kc, vc, cache_len, B, n_head, head_dim, max_len, scratch, stream)) return rc;
    const int nchunk = (max_len + LM_CHUNK - 1) / LM_CHUNK;
    hipLaunchKernelGGL(lm_attn_merge_kernel, dim3(n_head, B), dim3(128), 0, stream, scratch, cache_len, n_head, head_dim,
                       nchunk, 0, out, LM_CHUNK);
    OT_LAUNCH_CHECK("lm_attn_merge");
    return OMNITOK_OK;
}

// ---------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------
struct LmLayer {
    const float *ln1w, *ln1b, *ln2w, *ln2b;
    float *wqkv, *bqkv;  // (query | key | value) concatenated at finalize
    const float *wproj, *bproj, *w1, *b1, *w2, *b2;
};

struct omnitok_lm {
    omnitok_lm_config cfg;
    std::map<std::string, std::vector<int64_t>> spec;
    std::map<std::string, float *> w;
    std::vector<void *> owned;
    std::vector<LmLayer> layers;
    bool finalized = false;
    // cache + workspaces
    float *kv = nullptr;
    int max_batch = 0, max_len = 0;
    int chunk = LM_CHUNK;  // keys per attention chunk of the decode step (omnitok_lm_alloc_cache)
    int *err_flag = nullptr;  // set by the attention kernel when a stream steps past max_len
    float *x = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *part = nullptr;
    int64_t cache_bytes = 0;
    // grow-only prefill workspace (rows = B * T)
    float *pf = nullptr;
    int64_t pf_floats = 0;
};

static const float *LW(omnitok_lm *lm, const std::string &k) {
    auto it = lm->w.find(k);
    return it == lm->w.end() ? nullptr : it->second;
}

extern "C" int omnitok_lm_create(const omnitok_lm_config *cfg, omnitok_lm **out) {
    OT_CHECK_ARG(cfg && out, "lm_create: null pointer");
    const omnitok_lm_config &c = *cfg;
    OT_CHECK_ARG(c.n_layer > 0 && c.n_head > 0 && c.n_embd > 0 && c.vocab_size > 0 && c.block_size > 0,
                 "lm_create: bad config");
    const int hd = c.n_embd / c.n_head;
    if (c.n_embd % c.n_head || (hd != 64 && hd != 96 && hd != 128) || c.n_embd % 256) {
        set_error("lm_create: n_embd %d / n_head %d: head_dim must be 64, 96 or 128 and n_embd %% 256 == 0", c.n_embd,
                  c.n_head);
        return OMNITOK_ERR_UNSUPPORTED;
    }
    omnitok_lm *lm = new omnitok_lm();
    lm->cfg = c;
    const int64_t C = c.n_embd, V = c.vocab_size;
    lm->spec["pos_emb"] = {1, c.block_size, C};
    lm->spec["tok_emb.weight"] = {V, C};
    for (int i = 0; i < c.n_layer; ++i) {
        const std::string p = "blocks." + std::to_string(i);
        for (const char *n : {".ln1", ".ln2"}) {
            lm->spec[p + n + ".weight"] = {C};
            lm->spec[p + n + ".bias"] = {C};
        }
        for (const char *n : {".attn.key", ".attn.query", ".attn.value", ".attn.proj"}) {
            lm->spec[p + n + ".weight"] = {C, C};
            lm->spec[p + n + ".bias"] = {C};
        }
        lm->spec[p + ".mlp.0.weight"] = {4 * C, C};
        lm->spec[p + ".mlp.0.bias"] = {4 * C};
        lm->spec[p + ".mlp.2.weight"] = {C, 4 * C};
        lm->spec[p + ".mlp.2.bias"] = {C};
    }
    lm->spec["ln_f.weight"] = {C};
    lm->spec["ln_f.bias"] = {C};
    lm->spec["head.weight"] = {V, C};
    *out = lm;
    return OMNITOK_OK;
}

static void lm_free_cache(omnitok_lm *lm) {
    lm->pf_floats = 0;
    for (float **p : {&lm->kv, &lm->x, &lm->qkv, &lm->att, &lm->hid, &lm->part, &lm->pf})
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
}

extern "C" void omnitok_lm_destroy(omnitok_lm *lm) {
    if (!lm) return;
    for (auto &kv : lm->w)
        if (kv.second) (void)hipFree(kv.second);
    for (void *p : lm->owned) (void)hipFree(p);
    lm_free_cache(lm);
    if (lm->err_flag) (void)hipFree(lm->err_flag);
    delete lm;
}

extern "C" int omnitok_lm_set_weight(omnitok_lm *lm, const char *name, const void *dev_ptr, const int64_t *shape,
                                     int ndim, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(lm && name && dev_ptr && shape, "lm_set_weight: null pointer");
    auto it = lm->spec.find(name);
    if (it == lm->spec.end()) return 1;
    std::vector<int64_t> shp(shape, shape + ndim);
    if (shp != it->second) {
        std::string want, got;
        for (auto s : it->second) want += std::to_string(s) + ",";
        for (auto s : shp) got += std::to_string(s) + ",";
        set_error("lm_set_weight: size mismatch for %s: expected [%s] got [%s]", name, want.c_str(), got.c_str());
        return OMNITOK_ERR_INVALID;
    }
    int64_t n = 1;
    for (auto s : shp) n *= s;
    float *&p = lm->w[name];
    if (!p) OT_HIP(hipMalloc(reinterpret_cast<void **>(&p), (size_t)n * 4));
    OT_HIP(hipMemcpyAsync(p, dev_ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    lm->finalized = false;
    return OMNITOK_OK;
}

extern "C" int omnitok_lm_finalize(omnitok_lm *lm, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(lm, "lm_finalize: null");
    for (auto &kv : lm->spec)
        if (!lm->w.count(kv.first)) {
            set_error("lm_finalize: missing weight %s", kv.first.c_str());
            return OMNITOK_ERR_STATE;
        }
    for (auto &kv : lm->w)
        if (!kv.second) {  // q/k/v were folded into wqkv by an earlier finalize and released
            set_error("lm_finalize: %s was released by a previous finalize; set all weights again", kv.first.c_str());
            return OMNITOK_ERR_STATE;
        }
    OT_HIP(hipStreamSynchronize(stream));
    for (void *p : lm->owned) (void)hipFree(p);
    lm->owned.clear();
    lm->layers.clear();
    const int64_t C = lm->cfg.n_embd;
    for (int i = 0; i < lm->cfg.n_layer; ++i) {
        const std::string p = "blocks." + std::to_string(i);
        LmLayer L;
        L.ln1w = LW(lm, p + ".ln1.weight");
        L.ln1b = LW(lm, p + ".ln1.bias");
        L.ln2w = LW(lm, p + ".ln2.weight");
        L.ln2b = LW(lm, p + ".ln2.bias");
        void *wq = nullptr, *bq = nullptr;
        OT_HIP(hipMalloc(&wq, (size_t)3 * C * C * 4));
        OT_HIP(hipMalloc(&bq, (size_t)3 * C * 4));
        lm->owned.push_back(wq);
        lm->owned.push_back(bq);
        L.wqkv = static_cast<float *>(wq);
        L.bqkv = static_cast<float *>(bq);
        hipLaunchKernelGGL(lm_concat3_kernel, dim3((unsigned)((3 * C * C + 255) / 256)), dim3(256), 0, stream,
                           LW(lm, p + ".attn.query.weight"), LW(lm, p + ".attn.key.weight"),
                           LW(lm, p + ".attn.value.weight"), C * C, L.wqkv);
        hipLaunchKernelGGL(lm_concat3_kernel, dim3((unsigned)((3 * C + 255) / 256)), dim3(256), 0, stream,
                           LW(lm, p + ".attn.query.bias"), LW(lm, p + ".attn.key.bias"), LW(lm, p + ".attn.value.bias"),
                           C, L.bqkv);
        OT_LAUNCH_CHECK("lm_concat3");
        L.wproj = LW(lm, p + ".attn.proj.weight");
        L.bproj = LW(lm, p + ".attn.proj.bias");
        L.w1 = LW(lm, p + ".mlp.0.weight");
        L.b1 = LW(lm, p + ".mlp.0.bias");
        L.w2 = LW(lm, p + ".mlp.2.weight");
        L.b2 = LW(lm, p + ".mlp.2.bias");
        lm->layers.push_back(L);
    }
    // the separate q/k/v copies are no longer needed: free them (2.7 GB model -> no duplicate)
    OT_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < lm->cfg.n_layer; ++i)
        for (const char *n : {".attn.key", ".attn.query", ".attn.value"})
            for (const char *s : {".weight", ".bias"}) {
                const std::string k = "blocks." + std::to_string(i) + n + s;
                // keep the map entry (finalize checks presence) but release the memory
                if (lm->w[k]) {
                    (void)hipFree(lm->w[k]);
                    lm->w[k] = nullptr;
                }
            }
    lm->finalized = true;
    return OMNITOK_OK;
}

extern "C" int omnitok_lm_alloc_cache(omnitok_lm *lm, int max_batch, int max_len) {
    OT_CHECK_ARG(lm && max_batch > 0 && max_batch <= 16 && max_len > 0, "lm_alloc_cache: bad sizes (max_batch <= 16)");
    OT_CHECK_ARG(max_len <= LM_CHUNK * LM_MAX_CHUNKS && lm->cfg.n_head <= LM_MAX_HEADS,
                 "lm_alloc_cache: max_len <= %d and n_head <= %d", LM_CHUNK * LM_MAX_CHUNKS, LM_MAX_HEADS);
    lm_free_cache(lm);
    const omnitok_lm_config &c = lm->cfg;
    const int64_t C = c.n_embd;
    const int hd = c.n_embd / c.n_head;
    const int64_t per_layer = (int64_t)max_batch * c.n_head * max_len * hd;  // floats per K (or V)
    lm->chunk = g_lm_attn_short && max_len <= LM_CHUNK_SHORT * LM_MAX_CHUNKS ? LM_CHUNK_SHORT : LM_CHUNK;
    const int nchunk = (max_len + lm->chunk - 1) / lm->chunk;
    auto alloc = [](float **p, int64_t n) { return hipMalloc(reinterpret_cast<void **>(p), (size_t)n * 4); };
    OT_HIP(alloc(&lm->kv, per_layer * 2 * c.n_layer));
    OT_HIP(alloc(&lm->x, max_batch * C));
    OT_HIP(alloc(&lm->qkv, max_batch * 3 * C));
    OT_HIP(alloc(&lm->att, max_batch * C));
    OT_HIP(alloc(&lm->hid, max_batch * 4 * C));
    OT_HIP(alloc(&lm->part, (int64_t)max_batch * c.n_head * nchunk * (2 + hd)));
    lm->max_batch = max_batch;
    lm->max_len = max_len;
    lm->cache_bytes = per_layer * 2 * c.n_layer * 4;
    if (!lm->err_flag) {
        OT_HIP(hipMalloc(reinterpret_cast<void **>(&lm->err_flag), sizeof(int)));
        OT_HIP(hipMemset(lm->err_flag, 0, sizeof(int)));
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_lm_overflowed(omnitok_lm *lm, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(lm, "lm_overflowed: null engine");
    if (!lm->err_flag) return 0;
    int h = 0;
    OT_HIP(hipMemcpyAsync(&h, lm->err_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    OT_HIP(hipStreamSynchronize(stream));
    if (h)
        if (int rc = device_fill_u32(lm->err_flag, 0u, 1, stream)) return rc;
    return h ? 1 : 0;
}

extern "C" int64_t omnitok_lm_cache_bytes(omnitok_lm *lm) { return lm ? lm->cache_bytes : 0; }

extern "C" int omnitok_lm_step(omnitok_lm *lm, const int64_t *idx, int32_t *pos, int32_t *cache_len, int B,
                               float *logits_out, int advance, omnitok_stream_t stream_) {
    return omnitok_lm_step_ex(lm, idx, nullptr, nullptr, pos, cache_len, B, logits_out, advance, stream_);
}

extern "C" int omnitok_lm_step_ex(omnitok_lm *lm, const int64_t *idx, const float *emb, const float *pos_extra,
                                  int32_t *pos, int32_t *cache_len, int B, float *logits_out, int advance,
                                  omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(lm, "lm_step: null engine");
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG((idx || emb) && pos && cache_len, "lm_step: null pointer");
    if (!lm->finalized) {
        set_error("lm_step: engine not finalised (load the weights first)");
        return OMNITOK_ERR_STATE;
    }
    if (!lm->kv || B > lm->max_batch) {
        set_error("lm_step: cache not allocated for batch %d (omnitok_lm_alloc_cache)", B);
        return OMNITOK_ERR_STATE;
    }
    const omnitok_lm_config &c = lm->cfg;
    const int C = c.n_embd, hd = C / c.n_head;
    const int64_t per_layer = (int64_t)lm->max_batch * c.n_head * lm->max_len * hd;
    hipLaunchKernelGGL(lm_embed_kernel, dim3(B), dim3(256), 0, stream, idx, pos, LW(lm, "tok_emb.weight"),
                       LW(lm, "pos_emb"), emb, pos_extra, lm->x, C, c.vocab_size, c.block_size);
    OT_LAUNCH_CHECK("lm_embed");
    for (int i = 0; i < c.n_layer; ++i) {
        const LmLayer &L = lm->layers[i];
        float *kc = lm->kv + (int64_t)(2 * i) * per_layer, *vc = kc + per_layer;
        // x + proj(attn(ln1(x)))   (reference gpt.py:159-161)
        if (int rc = omnitok_lm_gemv(lm->x, L.wqkv, L.bqkv, nullptr, L.ln1w, L.ln1b, lm->qkv, B, 3 * C, C, 0, stream))
            return rc;
        if (int rc = lm_attn_partials(lm->qkv, kc, vc, cache_len, B, c.n_head, hd, lm->max_len, lm->part, stream, 0,
                                      lm->err_flag, lm->chunk))
            return rc;
        const LmMerge mg{lm->part, cache_len, c.n_head, hd, (lm->max_len + lm->chunk - 1) / lm->chunk, lm->chunk, lm->qkv};   // (qkv is free once the partials exist)
        if (int rc = lm_gemv_any(nullptr, L.wproj, L.bproj, lm->x, nullptr, nullptr, lm->x, B, C, C, 0, &mg, stream))
            return rc;
        // x + mlp(ln2(x))          (reference gpt.py:162, 150-155)
        if (int rc = omnitok_lm_gemv(lm->x, L.w1, L.b1, nullptr, L.ln2w, L.ln2b, lm->hid, B, 4 * C, C, 1, stream))
            return rc;
        if (int rc = omnitok_lm_gemv(lm->hid, L.w2, L.b2, lm->x, nullptr, nullptr, lm->x, B, C, 4 * C, 0, stream))
            return rc;
    }
    if (logits_out)  // ln_f + head (no bias), reference gpt.py:263-264
        if (int rc = omnitok_lm_gemv(lm->x, LW(lm, "head.weight"), nullptr, nullptr, LW(lm, "ln_f.weight"),
                                     LW(lm, "ln_f.bias"), logits_out, B, c.vocab_size, C, 0, stream))
            return rc;
    if (advance) {
        hipLaunchKernelGGL(lm_advance_kernel, dim3(1), dim3(64), 0, stream, pos, cache_len, B);
        OT_LAUNCH_CHECK("lm_advance");
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_lm_prefill(omnitok_lm *lm, const int64_t *idx, int32_t *pos, int32_t *cache_len, int B, int T,
                                  float *logits_out, omnitok_stream_t stream_) {
    return omnitok_lm_prefill_ex(lm, idx, T, nullptr, 0, nullptr, pos, cache_len, B, logits_out, stream_);
}

extern "C" int omnitok_lm_prefill_ex(omnitok_lm *lm, const int64_t *idx, int T_tok, const float *emb, int T_emb,
                                     const float *pos_extra, int32_t *pos, int32_t *cache_len, int B,
                                     float *logits_out, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(lm, "lm_prefill: null engine");
    OT_CHECK_ARG(T_tok >= 0 && T_emb >= 0, "lm_prefill: negative length");
    const int T = T_tok + T_emb;
    if (B == 0 || T == 0) return OMNITOK_OK;
    OT_CHECK_ARG((idx || T_tok == 0) && (emb || T_emb == 0) && pos && cache_len, "lm_prefill: null pointer");
    if (!lm->finalized) {
        set_error("lm_prefill: engine not finalised (load the weights first)");
        return OMNITOK_ERR_STATE;
    }
    if (!lm->kv || B > lm->max_batch || T > lm->max_len) {
        set_error("lm_prefill: cache not allocated for %d streams x %d tokens (omnitok_lm_alloc_cache)", B, T);
        return OMNITOK_ERR_STATE;
    }
    const omnitok_lm_config &c = lm->cfg;
    OT_CHECK_ARG(T <= c.block_size, "lm_prefill: %d tokens exceed block_size %d", T, c.block_size);
    const int C = c.n_embd, hd = C / c.n_head, V = c.vocab_size;
    const int64_t M = (int64_t)B * T;
    OT_CHECK_ARG(M <= 65535, "lm_prefill: B * T = %lld rows (max 65535)", (long long)M);
    const int nchunk = (T + LM_CHUNK - 1) / LM_CHUNK;
    // workspace: x | xn | att [M, C], qkv [M, 3C], hid [M, 4C], partials
    const int64_t need = M * C * 3 + M * 3 * C + M * 4 * C + M * c.n_head * nchunk * (2 + hd);
    if (lm->pf_floats < need) {
        if (lm->pf) OT_HIP(hipFree(lm->pf));
        lm->pf = nullptr;
        lm->pf_floats = 0;
        OT_HIP(hipMalloc(reinterpret_cast<void **>(&lm->pf), (size_t)need * 4));
        lm->pf_floats = need;
    }
    float *x = lm->pf, *xn = x + M * C, *att = xn + M * C, *qkv = att + M * C, *hid = qkv + M * 3 * C,
          *part = hid + M * 4 * C;
    const int64_t per_layer = (int64_t)lm->max_batch * c.n_head * lm->max_len * hd;
    const dim3 ln_grid((unsigned)((M + 3) / 4));
    hipLaunchKernelGGL(lm_embed_seq_kernel, dim3((unsigned)M), dim3(256), 0, stream, idx, LW(lm, "tok_emb.weight"),
                       LW(lm, "pos_emb"), emb, T_emb, pos_extra, x, T, C, V);
    OT_LAUNCH_CHECK("lm_embed_seq");
    const int BR = OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL;
    for (int i = 0; i < c.n_layer; ++i) {
        const LmLayer &L = lm->layers[i];
        float *kc = lm->kv + (int64_t)(2 * i) * per_layer, *vc = kc + per_layer;
        hipLaunchKernelGGL(lm_layernorm_rows_kernel, ln_grid, dim3(256), 0, stream, x, L.ln1w, L.ln1b, xn, M, C);
        OT_LAUNCH_CHECK("lm_layernorm_rows");
        if (int rc = omnitok_gemm(xn, C, L.wqkv, C, L.bqkv, nullptr, 0, qkv, 3 * C, M, 3 * C, C, OMNITOK_GEMM_BIAS, 0, 0,
                                  0, stream))
            return rc;
        hipLaunchKernelGGL(lm_kv_scatter_kernel, dim3((unsigned)M), dim3(256), 0, stream, qkv, kc, vc, T, c.n_head, hd,
                           lm->max_len);
        OT_LAUNCH_CHECK("lm_kv_scatter");
        if (int rc = lm_attn_partials(qkv, kc, vc, nullptr, B, c.n_head, hd, lm->max_len, part, stream, T)) return rc;
        hipLaunchKernelGGL(lm_attn_merge_kernel, dim3(c.n_head, (unsigned)M), dim3(128), 0, stream, part, nullptr,
                           c.n_head, hd, nchunk, T, att, LM_CHUNK);
        OT_LAUNCH_CHECK("lm_attn_merge");
        if (int rc = omnitok_gemm(att, C, L.wproj, C, L.bproj, x, C, x, C, M, C, C, BR, 0, 0, 0, stream)) return rc;
        hipLaunchKernelGGL(lm_layernorm_rows_kernel, ln_grid, dim3(256), 0, stream, x, L.ln2w, L.ln2b, xn, M, C);
        OT_LAUNCH_CHECK("lm_layernorm_rows");
        if (int rc = omnitok_gemm(xn, C, L.w1, C, L.b1, nullptr, 0, hid, 4 * C, M, 4 * C, C, OMNITOK_GEMM_BIAS, 0, 0, 0,
                                  stream))
            return rc;
        hipLaunchKernelGGL(lm_gelu_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, stream, hid, M * C);
        OT_LAUNCH_CHECK("lm_gelu");
        if (int rc = omnitok_gemm(hid, 4 * C, L.w2, 4 * C, L.b2, x, C, x, C, M, C, 4 * C, BR, 0, 0, 0, stream)) return rc;
    }
    if (logits_out) {
        hipLaunchKernelGGL(lm_layernorm_rows_kernel, ln_grid, dim3(256), 0, stream, x, LW(lm, "ln_f.weight"),
                           LW(lm, "ln_f.bias"), xn, M, C);
        OT_LAUNCH_CHECK("lm_layernorm_rows");
        if (int rc = omnitok_gemm(xn, C, LW(lm, "head.weight"), C, nullptr, nullptr, 0, logits_out, V, M, V, C, 0, 0, 0, 0,
                                  stream))
            return rc;
    }
    hipLaunchKernelGGL(lm_set_len_kernel, dim3(1), dim3(64), 0, stream, pos, cache_len, B, T);
    OT_LAUNCH_CHECK("lm_set_len");
    return OMNITOK_OK;
}
