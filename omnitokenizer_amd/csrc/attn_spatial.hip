// Spatial attention on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950:
//   * qk_prep        -- RoPE(2-D) + l2norm + q/k scale, in place   (reference attention.py:417-437)
//   * attn_spatial   -- full attention over N tokens, flash style  (attention.py:439-483)
//   * attn_window    -- 8x8 window attention with rel-pos bias     (attention.py:254-293)
//
// MFMA formulation (both kernels).  Everything is computed TRANSPOSED so that all per-query
// statistics are lane-local and P feeds the second MFMA without any cross-lane movement:
//     S^T[kv][q] = K . Q^T        A = K fragment,  B = Q fragment      (32 MFMA per 32x32 tile)
//     O^T[d][q]  = V^T . P^T      A = V fragment,  B = P = exp(S^T-m)  (32 MFMA per 32 kv x 64 d)
// In the 32x32 C/D layout lane l owns column q = l&31 and rows (r&3)+8*(r>>2)+4*(l>>5); the B
// operand of 32x32x2 wants from lane l the element [k = l>>5][j = l&31] -- exactly what the lane
// already holds in accumulator register r when the two k-slots of that MFMA step are defined to be
// kv = (r&3)+8*(r>>2) and kv+4.  So P is consumed straight from the S^T accumulators, the running
// max / sum / rescale are per-lane scalars, and only one exchange with lane^32 per reduction is
// needed.  The d (=64) contraction of S^T is split lane-wise: lanes 0-31 contribute d in [0,32),
// lanes 32-63 d in [32,64), so every Q/K fragment is 32 contiguous floats (8 x 16-byte reads).
// fp32 throughout: the reference computes this path in fp32 and the parity bar (ids bit-exact,
// pixels 1e-4) does not survive bf16 (SURVEY.md section 7 "Hard parts").
#include "common.h"

#include "planes.h"

namespace omnitok {

// exp(x) for x <= 0 as one multiply-add and one v_exp_f32 (1 ulp): on gfx950 every VALU instruction
// issued between fp32 MFMAs costs matrix-pipe time (measured ~5 cycles each), and the library expf
// is ~15 instructions per element of P.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// -------------------------------------------------------------------------------------------
// qk_prep: one 16-lane row per (token, head); each lane owns 4 consecutive channels, so the RoPE
// pairs (2j, 2j+1) are lane-local and the l2 norm is a 4-step DPP all-reduce.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_prep_kernel(float *__restrict__ q, int64_t ldq, float *__restrict__ k,
                                                      int64_t ldk, int64_t rows, int n_tokens, int heads,
                                                      const float *__restrict__ cosT,
                                                      const float *__restrict__ sinT,
                                                      const float *__restrict__ q_scale,
                                                      const float *__restrict__ k_scale, float scale) {
    const int l16 = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;  // (row, head)
    const int64_t ngrp = rows * heads;
    if (grp >= ngrp) return;  // whole 16-lane rows exit together
    const int64_t row = grp / heads;
    const int head = (int)(grp % heads);
    const int n = (int)(row % n_tokens);
    float c0 = 1.f, c1 = 1.f, s0 = 0.f, s1 = 0.f;
    if (cosT) {
        c0 = cosT[n * 32 + 2 * l16]; c1 = cosT[n * 32 + 2 * l16 + 1];
        s0 = sinT[n * 32 + 2 * l16]; s1 = sinT[n * 32 + 2 * l16 + 1];
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        float *base = which == 0 ? q + row * ldq : k + row * ldk;
        f32x4 *p = reinterpret_cast<f32x4 *>(base + head * 64 + l16 * 4);
        f32x4 v = *p;
        if (cosT) {  // (a + ib)(c + is) = (ac - bs) + i(as + bc), reference attention.py:65-69
            const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
            v[0] = a0 * c0 - b0 * s0; v[1] = a0 * s0 + b0 * c0;
            v[2] = a1 * c1 - b1 * s1; v[3] = a1 * s1 + b1 * c1;
        }
        float ss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        ss = row16_allsum(ss);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps, attention.py:24-25
        const f32x4 sc = *reinterpret_cast<const f32x4 *>((which == 0 ? q_scale : k_scale) + l16 * 4);
        const float mul = which == 0 ? scale : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * inv * sc[e] * mul;
        *p = v;
    }
}

// -------------------------------------------------------------------------------------------
// Full spatial attention.  Workgroup = 4 waves = 128 queries of one (sequence, head); each wave
// owns 32 queries.  K/V tiles of 64 keys are staged global -> registers -> LDS (double-buffered,
// next tile's loads in flight during the current tile's 128 MFMAs).
// -------------------------------------------------------------------------------------------
constexpr int AT_KV = 64;        // keys per LDS tile
constexpr int AT_LDK = 68;       // padded K row (floats): conflict-free ds_read_b128 fragments
constexpr int AT_LDV = 64;
constexpr int AT_STAGE_FLOATS = AT_KV * AT_LDK + AT_KV * AT_LDV;
constexpr int AT_LDS_BYTES = 2 * AT_STAGE_FLOATS * 4;

struct AttnParams {
    const float *q; const float *k; const float *v; float *out;
    int64_t ldq, ldkv, ldo;
    int N, heads, nqb, ngrp_real;
    const float *bias_table;  // [(2gh-1)*(2gw-1), heads] or null
    int gh, gw;
};

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hi = lane >> 5;
    // XCD-aware mapping (workgroup b runs on XCD b % 8, each XCD has its own L2): all query blocks
    // of one (sequence, head) are placed on the same XCD so that its K/V (256 KiB at N = 1024) is
    // fetched from HBM once and re-read from that L2 by the other query blocks -- measured without
    // this: 5.7 GB of memory-side reads per launch at C3 against 1.0 GB algorithmic.
    const int nqb = p.nqb;
    const int b = blockIdx.x, j = b >> 3;
    const int grp = (j / nqb) * 8 + (b & 7);  // (sequence, head) pair; heads * Bn is a multiple of 8
    if (grp >= p.ngrp_real) return;  // padding workgroup (uniform exit before any barrier)
    const int qb = j % nqb, head = grp % p.heads, seq = grp / p.heads;
    const int64_t seq_row0 = (int64_t)seq * p.N;
    // N % 64 == 0 but not necessarily % 128: the last workgroup may have idle waves, which still
    // take part in the cooperative K/V staging and barriers but store nothing
    const bool wave_active = qb * 128 + wave * 32 < p.N;
    const int q_local = wave_active ? qb * 128 + wave * 32 + r32 : r32;  // this lane's query

    // Q fragment: 32 contiguous floats of this lane's query row, d in [hi*32, hi*32+32)
    f32x4 qf[8];
    {
        const f32x4 *qp = reinterpret_cast<const f32x4 *>(p.q + (seq_row0 + q_local) * p.ldq + head * 64 + hi * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = qp[j];
    }
    int qy = 0, qx = 0;
    const float *btab = nullptr;
    if constexpr (HAS_BIAS) {
        qy = q_local / p.gw; qx = q_local % p.gw;
        btab = p.bias_table + head;
    }

    // loader mapping: 64 rows x 16 float4 per matrix; thread -> row tid/16 + 16*i, col4 tid%16
    const int lrow = tid >> 4, lc4 = tid & 15;
    const float *kbase = p.k + seq_row0 * p.ldkv + head * 64 + lc4 * 4;
    const float *vbase = p.v + seq_row0 * p.ldkv + head * 64 + lc4 * 4;
    f32x4 rk[4], rv[4];
    auto gload = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t off = (int64_t)(kv0 + lrow + 16 * i) * p.ldkv;
            rk[i] = *reinterpret_cast<const f32x4 *>(kbase + off);
            rv[i] = *reinterpret_cast<const f32x4 *>(vbase + off);
        }
    };
    auto lstore = [&](int buf) {
        float *Ks = smem + buf * AT_STAGE_FLOATS;
        float *Vs = Ks + AT_KV * AT_LDK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(Ks + (lrow + 16 * i) * AT_LDK + lc4 * 4) = rk[i];
            *reinterpret_cast<f32x4 *>(Vs + (lrow + 16 * i) * AT_LDV + lc4 * 4) = rv[i];
        }
    };

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int ntiles = p.N / AT_KV;
    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload((t + 1) * AT_KV);
        const float *Ks = smem + buf * AT_STAGE_FLOATS;
        const float *Vs = Ks + AT_KV * AT_LDK;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // ---- S^T = K . Q^T for 32 keys x 32 queries -----------------------------------
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
            const f32x4 *kp = reinterpret_cast<const f32x4 *>(Ks + (sub * 32 + r32) * AT_LDK + hi * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 kf = kp[j];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[j][e], st, 0, 0, 0);
            }
            if constexpr (HAS_BIAS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * AT_KV + sub * 32 + mfma32_row(r, hi);
                    const int ky = kv / p.gw, kx = kv % p.gw;
                    st[r] += btab[((qy - ky + p.gh - 1) * (2 * p.gw - 1) + (qx - kx + p.gw - 1)) * p.heads];
                }
            }
            // ---- online softmax (per-lane query) ------------------------------------------
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = fmaxf(mx, swap32(mx));
            const float m_new = fmaxf(m_run, mx);
            const float mc = m_new * 1.44269504088896340736f;
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(fmaf(st[r], 1.44269504088896340736f, -mc));
                ps += st[r];
            }
            ps += swap32(ps);
            // rescale only when some query of the wave saw its running max grow (wave-uniform
            // branch; exact: alpha == 1 for every lane otherwise)
            if (__any(m_new != m_run)) {
                const float alpha = fast_exp(m_run - m_new);  // 0 on the first tile (m_run = -inf)
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
            }
            l_run += ps;
            m_run = m_new;
            // ---- O^T += V^T . P^T -----------------------------------------------------------
            // step r: k-slot 0 -> key (r&3)+8*(r>>2), k-slot 1 -> that + 4 (lanes 32-63)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *vrow = Vs + (sub * 32 + mfma32_row(r, hi)) * AT_LDV + r32;
                ot[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], st[r], ot[0], 0, 0, 0);
                ot[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], st[r], ot[1], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane owns q, d = dblk*32 + 8g + 4hi + (0..3) -----
    if (!wave_active) return;
    const float inv_l = 1.0f / l_run;
    float *orow = p.out + (seq_row0 + q_local) * p.ldo + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ot[d][g * 4 + e] * inv_l;
            *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
        }
}

// -------------------------------------------------------------------------------------------
// Window attention: one wave per (window, head): 64 queries x 64 keys x 64 channels, operands
// straight from global memory (each K/V element is used by exactly one wave), no LDS.
// -------------------------------------------------------------------------------------------
struct WinParams {
    const float *qkv; int64_t ldqkv;
    const float *bias_dense;  // [heads][64 kv][64 q]
    float *out; int64_t ldo;
    int gh, gw, heads, nwin_x, nwin;  // windows per row / per image
    // optional plane output (A operand of the proj GEMM, gemm_pl.h) with one static power-of-two scale
    unsigned char *out_planes;
    float out_mul;  // h2 scale of the static bound of |attention output|
};

__device__ __forceinline__ int win_token(int wy, int wx, int idx, int gw) {
    return (wy * 8 + (idx >> 3)) * gw + wx * 8 + (idx & 7);
}

__global__ __launch_bounds__(256, 2) void attn_window_kernel(WinParams p, int64_t total_units) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (image, window, head)
    if (unit >= total_units) return;
    const int r32 = lane & 31, hi = lane >> 5;
    const int head = (int)(unit % p.heads);
    const int64_t wi = unit / p.heads;
    const int win = (int)(wi % p.nwin);
    const int64_t img = wi / p.nwin;
    const int wy = win / p.nwin_x, wx = win % p.nwin_x;
    const int64_t row0 = img * (int64_t)p.gh * p.gw;
    const int hd = p.heads * 64;

    // Q fragments for the two 32-query blocks
    f32x4 qf[2][8];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int tok = win_token(wy, wx, qb * 32 + r32, p.gw);
        const f32x4 *qp = reinterpret_cast<const f32x4 *>(p.qkv + (row0 + tok) * p.ldqkv + head * 64 + hi * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[qb][j] = qp[j];
    }
    // S^T[kb][qb]: 4 tiles
    f32x16 st[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int tok = win_token(wy, wx, kb * 32 + r32, p.gw);
        const f32x4 *kp =
            reinterpret_cast<const f32x4 *>(p.qkv + (row0 + tok) * p.ldqkv + hd + head * 64 + hi * 32);
        f32x4 kf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = kp[j];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][qb][r] = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st[kb][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j][e], qf[qb][j][e], st[kb][qb], 0, 0, 0);
        }
    }
    // softmax over the 64 keys of each query: logits = 0.125 * q.k + bias (attention.py:274-282)
    const float *bh = p.bias_dense + (int64_t)head * 64 * 64;
    float inv_l[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kb * 32 + mfma32_row(r, hi);
                const float s = st[kb][qb][r] * 0.125f + bh[kv * 64 + qb * 32 + r32];
                st[kb][qb][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, swap32(mx));
        float ps = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp(st[kb][qb][r] - mx);
                st[kb][qb][r] = e;
                ps += e;
            }
        ps += swap32(ps);
        inv_l[qb] = 1.0f / ps;
    }
    // O^T[dblk][qb] = V^T . P^T
    f32x16 ot[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][qb][r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tok = win_token(wy, wx, kb * 32 + mfma32_row(r, hi), p.gw);
            const float *vrow = p.qkv + (row0 + tok) * p.ldqkv + 2 * hd + head * 64 + r32;
            const float v0 = vrow[0], v1 = vrow[32];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                ot[0][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, st[kb][qb][r], ot[0][qb], 0, 0, 0);
                ot[1][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, st[kb][qb][r], ot[1][qb], 0, 0, 0);
            }
        }
    if (p.out_planes) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int64_t row = row0 + win_token(wy, wx, qb * 32 + r32, p.gw);
#pragma unroll
            for (int d = 0; d < 2; ++d)
                pl_store_ot_block(ot[d][qb], inv_l[qb] * p.out_mul, p.out_planes, row, head * 2 + d, p.heads * 2, hi);
        }
        return;
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int tok = win_token(wy, wx, qb * 32 + r32, p.gw);
        float *orow = p.out + (row0 + tok) * p.ldo + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ot[d][qb][g * 4 + e] * inv_l[qb];
                *reinterpret_cast<f32x4 *>(orow + d * 32 + g * 8 + hi * 4) = o;
            }
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_qk_prep(float *q, int64_t ldq, float *k, int64_t ldk, int64_t rows, int n_tokens, int heads,
                               const float *cos, const float *sin, const float *q_scale, const float *k_scale,
                               float scale, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(q && k && q_scale && k_scale, "qk_prep: null pointer");
    OT_CHECK_ARG((cos == nullptr) == (sin == nullptr), "qk_prep: cos/sin must both be given or both null");
    OT_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && aligned16(q) && aligned16(k), "qk_prep: unaligned");
    const int64_t threads = rows * heads * 16;
    if (threads == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(qk_prep_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, q, ldq, k, ldk,
                       rows, n_tokens, heads, cos, sin, q_scale, k_scale, scale);
    OT_LAUNCH_CHECK("qk_prep");
    return OMNITOK_OK;
}

extern "C" int omnitok_attn_spatial(const float *q, int64_t ldq, const float *k, const float *v, int64_t ldkv,
                                    float *out, int64_t ldo, int Bn, int N, int heads, const float *bias_table,
                                    int gh, int gw, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(q && k && v && out, "attn_spatial: null pointer");
    OT_CHECK_ARG(N % 64 == 0 && N > 0, "attn_spatial: N=%d must be a multiple of 64 tokens", N);
    OT_CHECK_ARG(!bias_table || gh * gw == N, "attn_spatial: bias grid %dx%d != N=%d", gh, gw, N);
    OT_CHECK_ARG(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) &&
                     aligned16(out), "attn_spatial: unaligned");
    OT_CHECK_ARG((int64_t)heads * Bn * ((N + 127) / 128) < (1ll << 31) - 8, "attn_spatial: grid too large");
    if (Bn == 0) return OMNITOK_OK;
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out; p.ldq = ldq; p.ldkv = ldkv; p.ldo = ldo; p.N = N; p.heads = heads;
    p.bias_table = bias_table; p.gh = gh; p.gw = gw;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_kernel<false>), AT_LDS_BYTES)) return rc;
    if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(attn_spatial_kernel<true>), AT_LDS_BYTES)) return rc;
    p.nqb = (N + 127) / 128;
    // groups (sequence, head) are dealt to the 8 XCDs round-robin; pad the group count to a multiple
    // of 8 with idle workgroups when heads * Bn is not one
    const int ngrp = ((heads * Bn + 7) / 8) * 8;
    p.ngrp_real = heads * Bn;
    dim3 grid((unsigned)((int64_t)ngrp * p.nqb));
    if (bias_table)
        hipLaunchKernelGGL(attn_spatial_kernel<true>, grid, dim3(256), AT_LDS_BYTES, stream, p);
    else
        hipLaunchKernelGGL(attn_spatial_kernel<false>, grid, dim3(256), AT_LDS_BYTES, stream, p);
    OT_LAUNCH_CHECK("attn_spatial");
    return OMNITOK_OK;
}

extern "C" int omnitok_attn_window(const float *qkv, int64_t ldqkv, const float *bias_dense, float *out,
                                   int64_t ldo, int Bn, int gh, int gw, int heads, omnitok_stream_t stream_) {
    return omnitok_attn_window_planes(qkv, ldqkv, bias_dense, out, ldo, nullptr, 0.0f, Bn, gh, gw, heads, stream_);
}

// out_planes != NULL: the output goes to fp16 hi|lo planes (K = heads * 64; the A operand of the proj GEMM, gemm_pl.h)
// scaled by the power of two of out_bound (>= max |output|; the consumer's a_scale_const = omnitok_pl_unscale(out_bound))
extern "C" int omnitok_attn_window_planes(const float *qkv, int64_t ldqkv, const float *bias_dense, float *out,
                                          int64_t ldo, void *out_planes, float out_bound, int Bn, int gh, int gw,
                                          int heads, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(qkv && bias_dense && (out || out_planes), "attn_window: null pointer");
    OT_CHECK_ARG(gh % 8 == 0 && gw % 8 == 0, "attn_window: grid %dx%d not divisible by the 8x8 window", gh, gw);
    OT_CHECK_ARG(ldqkv % 4 == 0 && (out_planes || ldo % 4 == 0) && aligned16(qkv) && (!out || aligned16(out)), "attn_window: unaligned");
    OT_CHECK_ARG(!out_planes || (out_bound > 0.0f && aligned16(out_planes)), "attn_window: plane output needs a bound");
    WinParams p;
    p.qkv = qkv; p.ldqkv = ldqkv; p.bias_dense = bias_dense; p.out = out; p.ldo = ldo;
    p.gh = gh; p.gw = gw; p.heads = heads; p.nwin_x = gw / 8; p.nwin = (gh / 8) * (gw / 8);
    p.out_planes = static_cast<unsigned char *>(out_planes);
    p.out_mul = out_planes ? h2_scale_of_bound(out_bound) : 1.0f;
    const int64_t total = (int64_t)Bn * p.nwin * heads;
    if (total == 0) return OMNITOK_OK;
    hipLaunchKernelGGL(attn_window_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, p, total);
    OT_LAUNCH_CHECK("attn_window");
    return OMNITOK_OK;
}
