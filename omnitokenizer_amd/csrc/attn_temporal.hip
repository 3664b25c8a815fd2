// Temporal attention (reference attention.py:402-486, is_spatial=False, causal): T' <= ~17 tokens
// per (pixel column, head), 1.25 flop/byte -> HBM-bound.  No MFMA: one 16-lane DPP row per
// (column, head), each lane owns 4 of the 64 channels (16-byte accesses, a wave covers 1 KiB of
// contiguous q/k/v per token), dot products are 4 FMAs + a 4-step DPP all-reduce.  l2norm, the
// learned q/k scales and the SDPA scale are applied in registers, so q/k/v are read exactly once
// and nothing but the output is written.
#include "common.h"
#include "planes.h"

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
    // PLANES variants: the output as fp16 hi|lo planes (A operand of the to_out GEMM, gemm_pl.h) scaled per clip by the
    // power of two of the V bound; a workgroup = 16 consecutive columns of ONE head, staged through LDS so that the
    // global stores are runs of 16 T consecutive rows x 16 bytes
    unsigned char *out_planes;
    float *out_scale;            // [rows]: the factor that undoes the scale
    float v_bound; const float *v_bound_dev; int v_bound_stride; int64_t cols_per_clip;
};

// PLANES: thread -> (column, head); stage one token's 4 output channels
template <bool PLANES>
__device__ __forceinline__ void tattn_map(const TAttnParams &p, int64_t &col, int &head, bool &live) {
    if constexpr (PLANES) {
        head = (int)(blockIdx.x % p.heads);
        col = (int64_t)(blockIdx.x / p.heads) * 16 + (threadIdx.x >> 4);
        live = true;  // cols % 16 == 0
    } else {
        const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
        live = grp < p.cols * p.heads;
        col = grp / p.heads;
        head = (int)(grp % p.heads);
    }
}

__device__ __forceinline__ float tattn_out_scale(const TAttnParams &p, int64_t col) {
    float b = p.v_bound;
    if (p.v_bound_dev) b *= p.v_bound_dev[(int64_t)p.v_bound_stride * (col / p.cols_per_clip)];
    return h2_scale_of_bound(b);
}

// LDS layout of the staged planes: [plane][chunk of 8 channels (8)][local row (16 T)][16 B]
__device__ __forceinline__ void tattn_stage(unsigned char *lds, int NR, int rl, int l16, f32x4 o) {
    const f16x4 h = __builtin_convertvector(o, f16x4);
    const f16x4 l = __builtin_convertvector(o - __builtin_convertvector(h, f32x4), f16x4);
    unsigned char *dst = lds + ((l16 >> 1) * NR + rl) * 16 + (l16 & 1) * 8;
    *reinterpret_cast<u32x2 *>(dst) = __builtin_bit_cast(u32x2, h);
    *reinterpret_cast<u32x2 *>(dst + 8 * NR * 16) = __builtin_bit_cast(u32x2, l);
}

__device__ __forceinline__ void tattn_flush(const TAttnParams &p, const unsigned char *lds, int NR, int64_t row0_wg, int head) {
    __syncthreads();
    const int kblocks = p.heads * 2;
    for (int sidx = threadIdx.x; sidx < 16 * NR; sidx += 256) {
        const int rl = sidx % NR, pc = sidx / NR;  // pc = plane * 8 + chunk
        const int chunk = pc & 7, plane = pc >> 3;
        const u32x4 w = *reinterpret_cast<const u32x4 *>(lds + (int64_t)sidx * 16);
        *reinterpret_cast<u32x4 *>(p.out_planes + pl_chunk_offset(row0_wg + rl, head * 2 + (chunk >> 2), chunk & 3, kblocks) +
                                   plane * 4096) = w;
    }
}

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
template <int TMAX, bool CAUSAL, bool PLANES = false>
__global__ __launch_bounds__(256) void attn_temporal_reg(TAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tattn_lds[];
    const int l16 = threadIdx.x & 15;
    int64_t col;
    int head;
    bool live;
    tattn_map<PLANES>(p, col, head, live);
    if (!live) return;
    const int T = p.T;
    const float so = PLANES ? tattn_out_scale(p, col) : 1.0f;
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
        if constexpr (PLANES) {
            if (i < T) {
                tattn_stage(tattn_lds, 16 * T, (int)(threadIdx.x >> 4) * T + i, l16, o * (1.0f / l) * so);
                if (head == 0 && l16 == 0) p.out_scale[row0 + i] = 1.0f / so;
            }
        } else {
            if (i < T) *reinterpret_cast<f32x4 *>(p.out + (row0 + i) * p.ldo + choff) = o * (1.0f / l);
        }
    }
    if constexpr (PLANES) tattn_flush(p, tattn_lds, 16 * T, (col - (threadIdx.x >> 4)) * T, head);
}

// register-resident K / V with the queries streamed (one load in flight ahead of the one being used): the form for
// 9 < T <= TMAX (17 tokens = the 65-frame clips of BASELINE config C5), where holding q as well would not fit and
// the streaming variant below re-normalises every key for every query (1.4 TB/s at T = 17).
template <int TMAX, bool CAUSAL, bool PLANES = false>
__global__ __launch_bounds__(256) void attn_temporal_kv_reg(TAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tattn_lds[];
    const int l16 = threadIdx.x & 15;
    int64_t col;
    int head;
    bool live;
    tattn_map<PLANES>(p, col, head, live);
    if (!live) return;
    const int T = p.T;
    const float so = PLANES ? tattn_out_scale(p, col) : 1.0f;
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
        if constexpr (PLANES) {
            if (i < T) {
                tattn_stage(tattn_lds, 16 * T, (int)(threadIdx.x >> 4) * T + i, l16, o * (1.0f / l) * so);
                if (head == 0 && l16 == 0) p.out_scale[row0 + i] = 1.0f / so;
            }
        } else {
            if (i < T) *reinterpret_cast<f32x4 *>(p.out + (row0 + i) * p.ldo + choff) = o * (1.0f / l);
        }
    }
    if constexpr (PLANES) tattn_flush(p, tattn_lds, 16 * T, (col - (threadIdx.x >> 4)) * T, head);
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
    return omnitok_attn_temporal_planes(q, ldq, k, v, ldkv, out, ldo, nullptr, nullptr, 0.0f, nullptr, 1, 0, cols, T, heads,
                                        q_scale, k_scale, scale, causal, alibi_slopes, stream_);
}

// out_planes != NULL (T <= 17, cols % 16 == 0): the output goes to fp16 hi|lo planes (K = heads * 64) scaled per clip by the
// power of two of v_bound (x v_bound_dev[v_bound_stride * (col / cols_per_clip)]) >= max |v| >= max |output|;
// out_scale[row] receives the factor that undoes it.
extern "C" int omnitok_attn_temporal_planes(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv,
                                            float *out, int64_t ldo, void *out_planes, float *out_scale, float v_bound,
                                            const float *v_bound_dev, int v_bound_stride, int64_t cols_per_clip,
                                            int64_t cols, int T, int heads, const float *q_scale, const float *k_scale,
                                            float scale, int causal, const float *alibi_slopes, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(q && k && v && (out || out_planes) && q_scale && k_scale, "attn_temporal: null pointer");
    OT_CHECK_ARG(T > 0, "attn_temporal: T=%d", T);
    OT_CHECK_ARG(ldq % 4 == 0 && ldkv % 4 == 0 && (out_planes || ldo % 4 == 0) && aligned16(q) && aligned16(k) && aligned16(v) &&
                     (!out || aligned16(out)), "attn_temporal: unaligned");
    OT_CHECK_ARG(!out_planes || (out_scale && T <= 17 && cols % 16 == 0 && v_bound > 0.0f && aligned16(out_planes) &&
                                 (!v_bound_dev || (cols_per_clip > 0 && cols_per_clip % 16 == 0))),
                 "attn_temporal: plane output needs out_scale, T <= 17, cols %% 16 == 0 and a bound of |v|");
    TAttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out; p.ldq = ldq; p.ldkv = ldkv; p.ldo = ldo; p.cols = cols; p.T = T;
    p.heads = heads; p.q_scale = q_scale; p.k_scale = k_scale; p.scale = scale; p.causal = causal;
    p.alibi = alibi_slopes;
    p.out_planes = static_cast<unsigned char *>(out_planes); p.out_scale = out_scale; p.v_bound = v_bound;
    p.v_bound_dev = v_bound_dev; p.v_bound_stride = v_bound_stride; p.cols_per_clip = v_bound_dev ? cols_per_clip : (cols > 0 ? cols : 1);
    const int64_t threads = cols * heads * 16;
    if (threads == 0) return OMNITOK_OK;
    const dim3 grid((unsigned)((threads + 255) / 256));
    const int lds = out_planes ? 256 * T * 16 : 0;  // 2 planes x 8 chunks x 16 T rows x 16 B
#define OT_TLAUNCH(KERN, TM)                                                                         \
    do {                                                                                             \
        if (out_planes) {                                                                            \
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(KERN<TM, true, true>), 256 * TM * 16)) return rc;  \
            if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(KERN<TM, false, true>), 256 * TM * 16)) return rc; \
            if (causal)                                                                              \
                hipLaunchKernelGGL((KERN<TM, true, true>), grid, dim3(256), lds, stream, p);         \
            else                                                                                     \
                hipLaunchKernelGGL((KERN<TM, false, true>), grid, dim3(256), lds, stream, p);        \
        } else if (causal)                                                                           \
            hipLaunchKernelGGL((KERN<TM, true, false>), grid, dim3(256), 0, stream, p);              \
        else                                                                                         \
            hipLaunchKernelGGL((KERN<TM, false, false>), grid, dim3(256), 0, stream, p);             \
    } while (0)
    if (T <= 2)
        OT_TLAUNCH(attn_temporal_reg, 2);
    else if (T <= 5)
        OT_TLAUNCH(attn_temporal_reg, 5);
    else if (T <= 9)
        OT_TLAUNCH(attn_temporal_reg, 9);
    else if (T <= 17)
        OT_TLAUNCH(attn_temporal_kv_reg, 17);
    else
        hipLaunchKernelGGL(attn_temporal_stream, grid, dim3(256), 0, stream, p);
#undef OT_TLAUNCH
    OT_LAUNCH_CHECK("attn_temporal");
    return OMNITOK_OK;
}
