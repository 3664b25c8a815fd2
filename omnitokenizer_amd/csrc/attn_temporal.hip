// Temporal attention (reference attention.py:402-486, is_spatial=False, causal): T' <= ~17 tokens
// per (pixel column, head), 1.25 flop/byte -> HBM-bound.  No MFMA: one 16-lane DPP row per
// (column, head), each lane owns 4 of the 64 channels (16-byte accesses, a wave covers 1 KiB of
// contiguous q/k/v per token), dot products are 4 FMAs + a 4-step DPP all-reduce.  l2norm, the
// learned q/k scales and the SDPA scale are applied in registers, so q/k/v are read exactly once
// and nothing but the output is written.
#include "common.h"

namespace omnitok {

struct TAttnParams {
    const float *q; const float *k; const float *v; float *out;
    int64_t ldq, ldkv, ldo;
    int64_t cols;
    int T, heads;
    const float *q_scale; const float *k_scale;
    float scale;
    int causal;
    const float *alibi;  // [heads] or null
};

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) {
    return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
}

__device__ __forceinline__ f32x4 l2scale(f32x4 v, const f32x4 sc, float mul) {
    const float ss = row16_allsum(dot4(v, v));
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * inv * sc[e] * mul;
    return v;
}

// register-resident variant: all T tokens of k and v live in VGPRs (T <= TMAX), q is streamed
template <int TMAX, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_temporal_reg(TAttnParams p) {
    const int l16 = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (grp >= p.cols * p.heads) return;
    const int64_t col = grp / p.heads;
    const int head = (int)(grp % p.heads);
    const int T = p.T;
    const int64_t row0 = col * T;
    const int choff = head * 64 + l16 * 4;
    f32x4 q[TMAX], k[TMAX], v[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int tt = t < T ? t : T - 1;  // clamped loads keep the code branch-free
        q[t] = *reinterpret_cast<const f32x4 *>(p.q + (row0 + tt) * p.ldq + choff);
        k[t] = *reinterpret_cast<const f32x4 *>(p.k + (row0 + tt) * p.ldkv + choff);
        v[t] = *reinterpret_cast<const f32x4 *>(p.v + (row0 + tt) * p.ldkv + choff);
    }
    const f32x4 qs = *reinterpret_cast<const f32x4 *>(p.q_scale + l16 * 4);
    const f32x4 ks = *reinterpret_cast<const f32x4 *>(p.k_scale + l16 * 4);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) k[t] = l2scale(k[t], ks, 1.0f);
    const float slope = p.alibi ? p.alibi[head] : 0.0f;
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        const f32x4 qi = l2scale(q[i], qs, p.scale);
        float s[TMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            if (CAUSAL && j > i) continue;
            float d = row16_allsum(dot4(qi, k[j]));
            d -= slope * (float)(i > j ? i - j : j - i);  // ALiBi (legacy), attention.py:474
            d = j < T ? d : -INFINITY;                    // keys beyond T never contribute
            s[j] = d;
            mx = fmaxf(mx, d);
        }
        float l = 0.0f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            if (CAUSAL && j > i) continue;
            const float e = expf(s[j] - mx);  // exp(-inf) = 0 for j >= T
            l += e;
            o += v[j] * e;
        }
        if (i < T) *reinterpret_cast<f32x4 *>(p.out + (row0 + i) * p.ldo + choff) = o * (1.0f / l);
    }
}

// register-resident K / V with the queries streamed (one load in flight ahead of the one being used): the form for
// 9 < T <= TMAX (17 tokens = the 65-frame clips of BASELINE config C5), where holding q as well would not fit and
// the streaming variant below re-normalises every key for every query (1.4 TB/s at T = 17).
template <int TMAX, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_temporal_kv_reg(TAttnParams p) {
    const int l16 = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (grp >= p.cols * p.heads) return;
    const int64_t col = grp / p.heads;
    const int head = (int)(grp % p.heads);
    const int T = p.T;
    const int64_t row0 = col * T;
    const int choff = head * 64 + l16 * 4;
    f32x4 k[TMAX], v[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int tt = t < T ? t : T - 1;  // clamped loads keep the code branch-free
        k[t] = *reinterpret_cast<const f32x4 *>(p.k + (row0 + tt) * p.ldkv + choff);
        v[t] = *reinterpret_cast<const f32x4 *>(p.v + (row0 + tt) * p.ldkv + choff);
    }
    f32x4 q_next = *reinterpret_cast<const f32x4 *>(p.q + row0 * p.ldq + choff);
    const f32x4 qs = *reinterpret_cast<const f32x4 *>(p.q_scale + l16 * 4);
    const f32x4 ks = *reinterpret_cast<const f32x4 *>(p.k_scale + l16 * 4);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) k[t] = l2scale(k[t], ks, 1.0f);
    const float slope = p.alibi ? p.alibi[head] : 0.0f;
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        const f32x4 qi = l2scale(q_next, qs, p.scale);
        if (i + 1 < TMAX) {
            const int tn = i + 1 < T ? i + 1 : T - 1;
            q_next = *reinterpret_cast<const f32x4 *>(p.q + (row0 + tn) * p.ldq + choff);
        }
        float s[TMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            if (CAUSAL && j > i) continue;
            float d = row16_allsum(dot4(qi, k[j]));
            d -= slope * (float)(i > j ? i - j : j - i);  // ALiBi (legacy), attention.py:474
            d = j < T ? d : -INFINITY;                    // keys beyond T never contribute
            s[j] = d;
            mx = fmaxf(mx, d);
        }
        float l = 0.0f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            if (CAUSAL && j > i) continue;
            const float e = expf(s[j] - mx);  // exp(-inf) = 0 for j >= T
            l += e;
            o += v[j] * e;
        }
        if (i < T) *reinterpret_cast<f32x4 *>(p.out + (row0 + i) * p.ldo + choff) = o * (1.0f / l);
    }
}

// streaming variant for any T: per query an online-softmax sweep over the keys (K/V re-read from
// L1/L2; the column's working set is T * 12 KiB)
__global__ __launch_bounds__(256) void attn_temporal_stream(TAttnParams p) {
    const int l16 = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (grp >= p.cols * p.heads) return;
    const int64_t col = grp / p.heads;
    const int head = (int)(grp % p.heads);
    const int T = p.T;
    const int64_t row0 = col * T;
    const int choff = head * 64 + l16 * 4;
    const f32x4 qs = *reinterpret_cast<const f32x4 *>(p.q_scale + l16 * 4);
    const f32x4 ks = *reinterpret_cast<const f32x4 *>(p.k_scale + l16 * 4);
    const float slope = p.alibi ? p.alibi[head] : 0.0f;
    for (int i = 0; i < T; ++i) {
        const f32x4 qi = l2scale(*reinterpret_cast<const f32x4 *>(p.q + (row0 + i) * p.ldq + choff), qs, p.scale);
        float m = -INFINITY, l = 0.0f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        const int jend = p.causal ? i + 1 : T;
        for (int j = 0; j < jend; ++j) {
            const f32x4 kj = l2scale(*reinterpret_cast<const f32x4 *>(p.k + (row0 + j) * p.ldkv + choff), ks, 1.0f);
            const f32x4 vj = *reinterpret_cast<const f32x4 *>(p.v + (row0 + j) * p.ldkv + choff);
            float d = row16_allsum(dot4(qi, kj));
            d -= slope * (float)(i > j ? i - j : j - i);
            const float mn = fmaxf(m, d);
            const float alpha = expf(m - mn);
            const float e = expf(d - mn);
            l = l * alpha + e;
            o = o * alpha + vj * e;
            m = mn;
        }
        *reinterpret_cast<f32x4 *>(p.out + (row0 + i) * p.ldo + choff) = o * (1.0f / l);
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_attn_temporal(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv,
                                     float *out, int64_t ldo, int64_t cols, int T, int heads, const float *q_scale,
                                     const float *k_scale, float scale, int causal, const float *alibi_slopes,
                                     omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(q && k && v && out && q_scale && k_scale, "attn_temporal: null pointer");
    OT_CHECK_ARG(T > 0, "attn_temporal: T=%d", T);
    OT_CHECK_ARG(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) &&
                     aligned16(out), "attn_temporal: unaligned");
    TAttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out; p.ldq = ldq; p.ldkv = ldkv; p.ldo = ldo; p.cols = cols; p.T = T;
    p.heads = heads; p.q_scale = q_scale; p.k_scale = k_scale; p.scale = scale; p.causal = causal;
    p.alibi = alibi_slopes;
    const int64_t threads = cols * heads * 16;
    if (threads == 0) return OMNITOK_OK;
    const dim3 grid((unsigned)((threads + 255) / 256));
#define OT_TREG(TM)                                                                            \
    do {                                                                                       \
        if (causal)                                                                            \
            hipLaunchKernelGGL((attn_temporal_reg<TM, true>), grid, dim3(256), 0, stream, p);  \
        else                                                                                   \
            hipLaunchKernelGGL((attn_temporal_reg<TM, false>), grid, dim3(256), 0, stream, p); \
    } while (0)
    if (T <= 2)
        OT_TREG(2);
    else if (T <= 5)
        OT_TREG(5);
    else if (T <= 9)
        OT_TREG(9);
    else if (T <= 17) {
        if (causal)
            hipLaunchKernelGGL((attn_temporal_kv_reg<17, true>), grid, dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((attn_temporal_kv_reg<17, false>), grid, dim3(256), 0, stream, p);
    } else
        hipLaunchKernelGGL(attn_temporal_stream, grid, dim3(256), 0, stream, p);
#undef OT_TREG
    OT_LAUNCH_CHECK("attn_temporal");
    return OMNITOK_OK;
}
