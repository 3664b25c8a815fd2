// Token selection of the LM sampling loops as one kernel (reference OmniTokenizer/modules/gpt.py:347-357 and
// :428-441; `top_k_top_p_filtering`, gpt.py:19-51):
//     logits = logits[:, -1, :] / temperature                        [CFG: blend = (1 + t) lc - t lu of the two rows]
//     if top_k is not None: keep logits >= k-th largest (ties kept); then the nucleus: sorted descending, drop
//         every rank whose PREDECESSORS already hold more than top_p of the softmax mass (the first rank always stays)
//     probs = softmax(logits);  token = argmax (greedy)  |  one multinomial draw
// One workgroup per stream.  The draw is an inverse-CDF lookup with a uniform u[b] supplied by the caller (the
// host side takes it from torch's generator, so torch.manual_seed governs the samples): same distribution as
// torch.multinomial, different random-number consumption -- sampled tokens are distribution-equal, not
// stream-equal, to the reference (greedy decoding is token-equal).
//
// Sorted path (top_k >= 0): radix-select the k-th largest value (4 x 256-bin passes over order-preserving
// keys), compact the survivors into LDS as (key << 32 | ~index), bitonic sort (descending value, ascending
// index among equals: deterministic), exp / prefix-sum in rank order, nucleus cut, inverse CDF.
// Unsorted path (top_k < 0 = None): plain multinomial over the whole vocabulary in index order.
#include "common.h"

namespace omnitok {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_SORT_MAX = 16384;  // survivors the LDS sort buffer holds (128 KiB)
constexpr int SEL_VPT = 16;          // REG build: a thread keeps its V / 1024 values in registers (V <= 16384)

struct SelParams {
    const float *logits;    // [B, V]
    const float *logits_u;  // [B, V] unconditional rows (CFG) or null
    float temperature, c1, c2;
    int V, top_k, sample;
    float top_p;
    const float *u;     // [B] uniforms in [0, 1)
    int64_t *out;       // [B]
    float *blend_out;   // [B, V] or null: the logits the selection saw
    int *err_flag;      // set to 1 if the survivors do not fit the sort buffer
};

__device__ __forceinline__ unsigned sel_key(float d) {  // larger float -> larger key
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// value the selection sees for column i of row b: the reference's own sequence of roundings
// (logits / temperature, then (1 + t) * lc - t * lu with fp32 scalars, gpt.py:428-431)
__device__ __forceinline__ float sel_value(const SelParams &p, int64_t b, int i) {
    const float lc = __fdiv_rn(p.logits[b * p.V + i], p.temperature);
    if (!p.logits_u) return lc;
    const float lu = __fdiv_rn(p.logits_u[b * p.V + i], p.temperature);
    // __fmul_rn / __fsub_rn are plain operators in this HIP: each product is rounded on its own only behind no_fuse
    return no_fuse(p.c1 * lc) - no_fuse(p.c2 * lu);
}

// exclusive prefix sum of one float per thread over the workgroup (deterministic), total in *total
__device__ __forceinline__ float block_exclusive_scan(float v, float *scratch, float *total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) scratch[wave] = x;
    __syncthreads();
    if (wave == 0) {
        float w = lane < SEL_THREADS / 64 ? scratch[lane] : 0.0f;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const float y = __shfl_up(w, off);
            if (lane >= off) w += y;
        }
        if (lane < SEL_THREADS / 64) scratch[lane] = w;  // inclusive over waves
    }
    __syncthreads();
    const float base = wave ? scratch[wave - 1] : 0.0f;
    *total = scratch[SEL_THREADS / 64 - 1];
    __syncthreads();
    return base + x - v;
}

// REG: the row is read (and the CFG blend / temperature division done) once; every later pass runs on registers.
template <bool REG>
__global__ __launch_bounds__(SEL_THREADS) void lm_select_kernel(SelParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sel_sort[];  // [SEL_SORT_MAX] (sorted path)
    __shared__ float s_red[SEL_THREADS / 64];
    __shared__ int s_redi[SEL_THREADS / 64];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix, s_kleft, s_count;
    __shared__ float s_f[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const int V = p.V;

    // ---- pass 1: values (optionally written out), maximum and its first index -------------------------
    float mx = -INFINITY;
    int mi = 0x7FFFFFFF;
    float vals[REG ? SEL_VPT : 1];
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < SEL_VPT; ++j) {
            const int i = tid + j * SEL_THREADS;
            vals[j] = i < V ? sel_value(p, b, i) : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < SEL_VPT; ++j) {
            const int i = tid + j * SEL_THREADS;
            if (i < V) {
                if (p.blend_out) p.blend_out[b * V + i] = vals[j];
                if (vals[j] > mx) { mx = vals[j]; mi = i; }
            }
        }
    } else {
        for (int i = tid; i < V; i += SEL_THREADS) {
            const float v = sel_value(p, b, i);
            if (p.blend_out) p.blend_out[b * V + i] = v;
            if (v > mx) { mx = v; mi = i; }
        }
    }
    // f(value, column, valid) for this thread's columns, called by every lane of the wave the same number of times
    // (the wave-aggregated atomics below ballot inside f)
    auto each = [&](auto &&f) {
        if constexpr (REG) {
#pragma unroll
            for (int j = 0; j < SEL_VPT; ++j) {
                if (j * SEL_THREADS < V) {  // uniform
                    const int i = tid + j * SEL_THREADS;
                    f(vals[j], i, i < V);
                }
            }
        } else {
            for (int i0 = 0; i0 < V; i0 += SEL_THREADS) {
                const int i = i0 + tid;
                f(i < V ? sel_value(p, b, i) : -INFINITY, i, i < V);
            }
        }
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(mx, off);
        const int oi = __shfl_xor(mi, off);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    if (lane == 0) { s_red[wave] = mx; s_redi[wave] = mi; }
    __syncthreads();
    if (tid == 0) {
        float m = s_red[0];
        int ix = s_redi[0];
        for (int w = 1; w < SEL_THREADS / 64; ++w)
            if (s_red[w] > m || (s_red[w] == m && s_redi[w] < ix)) { m = s_red[w]; ix = s_redi[w]; }
        s_f[0] = m;
        s_redi[0] = ix == 0x7FFFFFFF ? 0 : ix;
    }
    __syncthreads();
    const float gmax = s_f[0];
    const int gidx = s_redi[0];
    __syncthreads();
    if (!p.sample) {  // torch.topk(probs, k=1): the filters never remove the maximum
        if (tid == 0) p.out[b] = gidx;
        return;
    }
    const float u = p.u[b];

    // ---- unsorted path: one inverse-CDF draw over the columns whose key is >= thr (index order) -----------------
    // (multinomial over all V when nothing is filtered; also the form of a top-k draw WITHOUT a nucleus cut when more
    // columns survive than the sort buffer holds -- top_k = 0 / top_k > 16384 on a large vocabulary, or massive ties at
    // the k-th value: no order is needed when top_p >= 1)
    auto unsorted_draw = [&](unsigned thr) {
        const int per = (V + SEL_THREADS - 1) / SEL_THREADS;
        const int i0 = tid * per, i1 = i0 + per < V ? i0 + per : V;
        auto mass = [&](int i) {
            const float v = sel_value(p, b, i);
            return sel_key(v) >= thr ? expf(v - gmax) : 0.0f;
        };
        float loc = 0.0f;
        for (int i = i0; i < i1; ++i) loc += mass(i);
        float total;
        const float before = block_exclusive_scan(loc, s_red, &total);
        const float target = u * total;
        if (tid == 0) s_count = 0x7FFFFFFF;
        __syncthreads();
        // the chunk that holds the target: before <= target < before + loc; ties of zero-mass chunks resolve to
        // the first non-empty one
        if (loc > 0.0f && before <= target && target < before + loc) {
            float c = before;
            int pick = -1;
            for (int i = i0; i < i1; ++i) {
                const float m = mass(i);
                c += m;
                if (m > 0.0f) pick = i;           // the last column with mass so far (rounding fell off the chunk's end)
                if (c > target && m > 0.0f) break;
            }
            if (pick >= 0) atomicMin(&s_count, (unsigned)pick);
        }
        __syncthreads();
        if (tid == 0) p.out[b] = s_count == 0x7FFFFFFFu ? gidx : (int64_t)s_count;  // rounding fell off the end: mode
    };
    if (p.top_k < 0) {
        unsorted_draw(0u);
        return;
    }

    // ---- sorted path ------------------------------------------------------------------------------------
    // k-th largest value by radix select on the order-preserving keys (top_k == 0: keep everything)
    unsigned thr_key = 0;  // keep keys >= thr_key
    if (p.top_k > 0 && p.top_k < V) {
        if (tid == 0) { s_prefix = 0; s_kleft = (unsigned)p.top_k; }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, pmask = pass ? 0xFFFFFFFFu << (shift + 8) : 0u;
            each([&](float v, int, bool valid) {
                const unsigned k = sel_key(v);
                const bool match = valid && (k & pmask) == prefix;
                const unsigned bin = (k >> shift) & 255;
                if (pass == 0) {
                    // sign + 7 exponent bits: a row of logits falls into a handful of bins, 8192 single atomics on
                    // them serialise; one atomic per (wave, distinct bin) instead
                    unsigned long long todo = __ballot(match);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned lb = __shfl(bin, leader);
                        const unsigned long long same = __ballot(match && bin == lb);
                        if (lane == leader) atomicAdd(&s_hist[lb], (unsigned)__popcll(same));
                        todo &= ~same;
                    }
                } else if (match) {
                    atomicAdd(&s_hist[bin], 1u);
                }
            });
            __syncthreads();
            if (wave == 0) {
                // the bin where the count from the top reaches k: the largest q with S_q = sum_{q' >= q} hist[q'] >= k.
                // One wave, four bins per lane, suffix sums by shuffles (a single thread walking the 256 bins was
                // 4 x ~10 us of dependent LDS reads: two thirds of the kernel).
                const unsigned left = s_kleft;
                const unsigned h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2],
                               h3 = s_hist[4 * lane + 3];
                const unsigned tot = (h0 + h1) + (h2 + h3);
                unsigned suf = tot;  // inclusive suffix sum over the lanes
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned y = __shfl_down(suf, off);
                    if (lane + off < 64) suf += y;
                }
                const unsigned above = suf - tot;  // keys in the bins of higher lanes
                const unsigned S3 = above + h3, S2 = S3 + h2, S1 = S2 + h1, S0 = S1 + h0;
                int bin = -1;
                unsigned over = 0;  // S_{bin + 1}
                if (S3 >= left && above < left) { bin = 4 * lane + 3; over = above; }
                else if (S2 >= left && S3 < left) { bin = 4 * lane + 2; over = S3; }
                else if (S1 >= left && S2 < left) { bin = 4 * lane + 1; over = S2; }
                else if (S0 >= left && S1 < left) { bin = 4 * lane; over = S1; }
                if (bin >= 0) {  // exactly one lane (1 <= left <= number of matching keys)
                    s_kleft = left - over;
                    s_prefix = prefix | ((unsigned)bin << shift);
                }
            }
            __syncthreads();
        }
        thr_key = s_prefix;
    }
    // compact the survivors (order irrelevant: the sort below is total)
    if (tid == 0) s_count = 0;
    __syncthreads();
    each([&](float v, int i, bool valid) {  // one counter atomic per wave and call
        const unsigned k = sel_key(v);
        const bool keep = valid && k >= thr_key;
        const unsigned long long m = __ballot(keep);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&s_count, (unsigned)__popcll(m));
            base = __shfl(base, leader);
            const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (keep && slot < SEL_SORT_MAX) sel_sort[slot] = ((unsigned long long)k << 32) | (0xFFFFFFFFu - (unsigned)i);
        }
    });
    __syncthreads();
    const int n_keep = (int)s_count;
    __syncthreads();  // s_count is reused below
    if (n_keep > SEL_SORT_MAX) {  // workgroup-uniform
        if (p.top_p >= 1.0f) {    // no nucleus cut: the draw needs no order
            unsorted_draw(thr_key);
            return;
        }
        if (tid == 0) {           // a nucleus over more survivors than the sort buffer: reported, never silently greedy
            if (p.err_flag) atomicOr(p.err_flag, 1);
            p.out[b] = gidx;
        }
        return;
    }
    int P = 128;
    while (P < n_keep) P <<= 1;
    for (int i = n_keep + tid; i < P; i += SEL_THREADS) sel_sort[i] = 0ull;  // padding sorts last
    __syncthreads();
    // bitonic sort, descending.  Compare-exchange distances j <= 64 stay inside aligned blocks of 128 elements: a wave
    // takes a block into registers (two elements per lane), does j = 64 between its two registers and j = 32 .. 1 by
    // shuffles, with no workgroup barrier; only the j >= 128 stages go through LDS stage by stage (P = 2048: 15
    // barriers instead of 66 -- with 16 waves a barrier costs about as much as a stage).
    auto inwave = [&](int k_lo, int k_hi) {  // phases k_lo .. k_hi, of each the sub-stages j = min(k / 2, 64) .. 1
        for (int blk = wave; blk < (P >> 7); blk += SEL_THREADS / 64) {
            const int e0 = (blk << 7) + lane, e1 = e0 + 64;
            unsigned long long x0 = sel_sort[e0], x1 = sel_sort[e1];
            for (int k = k_lo; k <= k_hi; k <<= 1) {
                int j = k >> 1;
                if (j >= 64) {
                    const bool desc = (e0 & k) == 0;
                    if ((x0 < x1) == desc) { const unsigned long long t = x0; x0 = x1; x1 = t; }
                    j = 32;
                }
                for (; j > 0; j >>= 1) {
                    const unsigned long long y0 = __shfl_xor(x0, j), y1 = __shfl_xor(x1, j);
                    const bool is_lo = (lane & j) == 0;
                    const bool d0 = ((e0 & ~j) & k) == 0, d1 = ((e1 & ~j) & k) == 0;  // direction of the pair
                    const unsigned long long mx0 = x0 > y0 ? x0 : y0, mn0 = x0 > y0 ? y0 : x0;
                    const unsigned long long mx1 = x1 > y1 ? x1 : y1, mn1 = x1 > y1 ? y1 : x1;
                    x0 = (is_lo == d0) ? mx0 : mn0;  // descending pair: the lower index takes the larger key
                    x1 = (is_lo == d1) ? mx1 : mn1;
                }
            }
            sel_sort[e0] = x0;
            sel_sort[e1] = x1;
        }
    };
    inwave(2, 128);
    __syncthreads();
    for (int k = 256; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 128; j >>= 1) {
            for (int t = tid; t < P / 2; t += SEL_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi2 = lo | j;
                const bool desc = (lo & k) == 0;
                const unsigned long long a = sel_sort[lo], c = sel_sort[hi2];
                if ((a < c) == desc) { sel_sort[lo] = c; sel_sort[hi2] = a; }
            }
            __syncthreads();
        }
        inwave(k, k);
        __syncthreads();
    }
    // probabilities in rank order (unnormalised), prefix sums over contiguous per-thread chunks
    const int per = (n_keep + SEL_THREADS - 1) / SEL_THREADS;
    const int r0 = tid * per < n_keep ? tid * per : n_keep, r1 = r0 + per < n_keep ? r0 + per : n_keep;
    auto prob = [&](int r) {
        const unsigned k = (unsigned)(sel_sort[r] >> 32);
        const unsigned uu = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
        return expf(__uint_as_float(uu) - gmax);
    };
    float loc = 0.0f;
    for (int r = r0; r < r1; ++r) loc += prob(r);
    float Z;
    const float before = block_exclusive_scan(loc, s_red, &Z);
    // nucleus: rank r stays iff r == 0 or cum_{r-1} <= top_p * Z  (a prefix); n_nuc = how many stay
    int n_nuc = n_keep;
    if (p.top_p < 1.0f) {
        const float lim = p.top_p * Z;
        if (tid == 0) s_count = 0;
        __syncthreads();
        unsigned cnt = 0;
        float c = before;
        for (int r = r0; r < r1; ++r) {
            if (r == 0 || c <= lim) ++cnt;   // c = cum_{r-1}
            c += prob(r);
        }
        if (cnt) atomicAdd(&s_count, cnt);
        __syncthreads();
        n_nuc = (int)s_count;
        if (n_nuc < 1) n_nuc = 1;
        __syncthreads();
    }
    // mass of the nucleus = cum_{n_nuc - 1}
    {
        if (tid == 0) s_f[1] = Z;
        __syncthreads();
        if (n_nuc < n_keep && r0 < n_nuc && n_nuc <= r1) {  // the chunk that holds rank n_nuc - 1
            float c = before;
            for (int r = r0; r < n_nuc; ++r) c += prob(r);
            s_f[1] = c;
        }
        __syncthreads();
    }
    const float target = u * s_f[1];
    if (tid == 0) s_count = 0x7FFFFFFF;
    __syncthreads();
    {
        const int e1 = r1 < n_nuc ? r1 : n_nuc;
        if (r0 < e1) {
            float c = before;
            for (int r = r0; r < e1; ++r) {
                c += prob(r);
                if (c > target) { atomicMin(&s_count, (unsigned)r); break; }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int r = s_count == 0x7FFFFFFFu ? n_nuc - 1 : (int)s_count;  // rounding fell off the end: last kept rank
        p.out[b] = (int64_t)(0xFFFFFFFFu - (unsigned)(sel_sort[r] & 0xFFFFFFFFull));
    }
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_lm_select(const float *logits, const float *logits_uncond, int B, int V, float temperature,
                                 float cfg_c1, float cfg_c2, int top_k, float top_p, int sample, const float *u,
                                 int64_t *out, float *blend_out, int *err_flag, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(logits && out, "lm_select: null pointer");
    OT_CHECK_ARG(V > 0 && B >= 0, "lm_select: bad shape");
    OT_CHECK_ARG(temperature > 0.0f, "lm_select: temperature must be positive");
    OT_CHECK_ARG(!sample || u, "lm_select: sampling needs the uniforms u[B]");
    OT_CHECK_ARG(top_p > 0.0f, "lm_select: top_p must be positive");
    if (B == 0) return OMNITOK_OK;
    SelParams p;
    p.logits = logits; p.logits_u = logits_uncond; p.temperature = temperature; p.c1 = cfg_c1; p.c2 = cfg_c2;
    p.V = V; p.top_k = top_k; p.sample = sample; p.top_p = top_p; p.u = u; p.out = out; p.blend_out = blend_out;
    p.err_flag = err_flag;
    const int lds = SEL_SORT_MAX * 8;
    if (V <= SEL_THREADS * SEL_VPT) {
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(lm_select_kernel<true>), lds)) return rc;
        hipLaunchKernelGGL(lm_select_kernel<true>, dim3(B), dim3(SEL_THREADS), lds, stream, p);
    } else {
        if (int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(lm_select_kernel<false>), lds)) return rc;
        hipLaunchKernelGGL(lm_select_kernel<false>, dim3(B), dim3(SEL_THREADS), lds, stream, p);
    }
    OT_LAUNCH_CHECK("lm_select");
    return OMNITOK_OK;
}
