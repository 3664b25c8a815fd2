// Engine, weight side: the state_dict contract (required keys and shapes), device copies, the repacked derivatives the
// kernels want (fp16 hi|lo weight planes for gemm_h2 / gemm_pl, GEGLU-interleaved FF weights, [27, D] PEG weights,
// A-fragment codebook, dense window bias, K-padded FF-out weight, BatchNorm folds), static operand ranges, RoPE and
// position-bias tables (reference omnitokenizer.py:806-838, 1006-1033; attention.py:28-70, 535-583).
#include "engine.h"

namespace omnitok {

Geo geometry(const omnitok_config &c) {
    Geo g;
    const bool lin = !c.patch_embed_cnn;  // the deferred pools exist for the 'linear' patch-embed only
    g.defer_t = lin && c.defer_temporal_pool;
    g.defer_s = lin && c.defer_spatial_pool;
    const int up = c.gen_upscale > 1 ? c.gen_upscale : 1;
    g.p_enc = g.defer_s ? c.patch_size / 2 : c.patch_size;
    g.pt_enc = g.defer_t ? c.temporal_patch_size / 2 : c.temporal_patch_size;
    g.p_dec = g.defer_s ? c.patch_size * up / 2 : c.patch_size * up;
    g.pt_dec = g.pt_enc;
    g.n_pool = 0;
    for (const char *q = c.enc_block; *q; ++q) g.n_pool += (*q == 'a' || *q == 'm' || *q == 'l');
    return g;
}

// token grid after the encoder's spatial blocks (pooling blocks halve it, attention.py:683-684; Up blocks
// double it, :686-687).  *peak (optional) = the largest tokens-per-frame count on the way (workspace size).
// false if a pooling block meets an odd grid.
bool walk_enc_grid(const char *block, int *gh, int *gw, int64_t *peak) {
    int64_t pk = (int64_t)*gh * *gw;
    for (const char *q = block; *q; ++q) {
        if (*q == 'a' || *q == 'm' || *q == 'l') {
            if (*gh % 2 || *gw % 2) return false;
            *gh /= 2;
            *gw /= 2;
        } else if (*q == 'n' || *q == 'r') {
            *gh *= 2;
            *gw *= 2;
        }
        if ((int64_t)*gh * *gw > pk) pk = (int64_t)*gh * *gw;
    }
    if (peak) *peak = pk;
    return true;
}

void add_transformer_spec(omnitok_engine *e, const std::string &prefix, const std::string &block,
                                 bool rel) {
    const omnitok_config &c = e->cfg;
    const int64_t d = c.dim, hd = c.dim_head, heads = c.heads, inner = c.ff_inner, ws = c.window_size;
    for (size_t i = 0; i < block.size(); ++i) {
        const std::string p = prefix + ".layers." + std::to_string(i);
        if (block[i] == 't') {
            e->spec[p + ".0.dsconv.weight"] = {d, 1, 3, 3, 3};
            e->spec[p + ".0.dsconv.bias"] = {d};
            e->spec[p + ".1.q_scale"] = {hd};
            e->spec[p + ".1.k_scale"] = {hd};
            if (rel) {
                e->spec[p + ".1.spatial_rel_pos_bias.net.0.0.weight"] = {d, 2};
                e->spec[p + ".1.spatial_rel_pos_bias.net.0.0.bias"] = {d};
                e->spec[p + ".1.spatial_rel_pos_bias.net.1.0.weight"] = {d, d};
                e->spec[p + ".1.spatial_rel_pos_bias.net.1.0.bias"] = {d};
                e->spec[p + ".1.spatial_rel_pos_bias.net.2.weight"] = {heads, d};
                e->spec[p + ".1.spatial_rel_pos_bias.net.2.bias"] = {heads};
            }
            e->spec[p + ".1.norm.gamma"] = {d};
            e->spec[p + ".1.norm.beta"] = {d};
            e->spec[p + ".1.to_q.weight"] = {hd * heads, d};
            e->spec[p + ".1.to_kv.weight"] = {2 * hd * heads, d};
            e->spec[p + ".1.to_out.weight"] = {d, hd * heads};
        } else if (block[i] == 'l') {
            e->spec[p + ".1.pool.weight"] = {d, 4 * d};
            e->spec[p + ".1.pool.bias"] = {d};
        } else if (block[i] == 'r') {  // Up('r'): Upsample -> Rearrange -> Linear, reference attention.py:122-127
            e->spec[p + ".1.up.2.weight"] = {d, d};
            e->spec[p + ".1.up.2.bias"] = {d};
        } else if (block[i] == 'a' || block[i] == 'm' || block[i] == 'n') {
            // parameter-free pooling / nearest up-sampling
        } else {
            e->spec[p + ".1.relative_position_bias_table"] = {(2 * ws - 1) * (2 * ws - 1), heads};
            e->spec[p + ".1.relative_position_index"] = {ws * ws, ws * ws};
            e->spec[p + ".1.norm.gamma"] = {d};
            e->spec[p + ".1.norm.beta"] = {d};
            e->spec[p + ".1.qkv.weight"] = {3 * d, d};
            e->spec[p + ".1.proj.weight"] = {d, d};
            e->spec[p + ".1.proj.bias"] = {d};
        }
        e->spec[p + ".3.0.weight"] = {d};
        e->spec[p + ".3.0.bias"] = {d};
        e->spec[p + ".3.1.weight"] = {2 * inner, d};
        e->spec[p + ".3.4.weight"] = {d, inner};
    }
    e->spec[prefix + ".norm_out.gamma"] = {d};
    e->spec[prefix + ".norm_out.beta"] = {d};
}

// keys that exist in the reference state_dict but are never read on the inference path
bool key_unused(const omnitok_engine *e, const std::string &k) {
    if (k.find("context_norm") != std::string::npos) return true;
    if (k == "codebook.N" || k == "codebook.z_avg" || k == "codebook.codebook_usage") return true;
    // temporal transformers never use their spatial_rel_pos_bias; spatial ones only
    // in legacy mode
    if (k.find("spatial_rel_pos_bias") != std::string::npos) {
        if (k.find("temporal_transformer") != std::string::npos) return true;
        return !e->cfg.legacy_attention;
    }
    return false;
}

int alloc_f(omnitok_engine *e, float **out, int64_t n) {
    void *p = nullptr;
    OT_HIP(hipMalloc(&p, (size_t)n * sizeof(float)));
    e->owned.push_back(p);
    *out = static_cast<float *>(p);
    return OMNITOK_OK;
}

int ensure(Buf &b, int64_t n) {
    if (b.cap >= n) return OMNITOK_OK;
    if (b.p) OT_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    void *p = nullptr;
    OT_HIP(hipMalloc(&p, (size_t)n * sizeof(float)));
    b.p = static_cast<float *>(p);
    b.cap = n;
    return OMNITOK_OK;
}

// quantiser weight names: Codebook + pre/post_vq_conv, or the external VectorQuantize's own projections
const char *k_embed(const omnitok_config &c) { return c.external_codebook ? "codebook._codebook.embed" : "codebook.embeddings"; }
const char *k_pre_w(const omnitok_config &c) { return c.external_codebook ? "codebook.project_in.weight" : "pre_vq_conv.1.weight"; }
const char *k_pre_b(const omnitok_config &c) { return c.external_codebook ? "codebook.project_in.bias" : "pre_vq_conv.1.bias"; }
const char *k_post_w(const omnitok_config &c) { return c.external_codebook ? "codebook.project_out.weight" : "post_vq_conv.1.weight"; }
const char *k_post_b(const omnitok_config &c) { return c.external_codebook ? "codebook.project_out.bias" : "post_vq_conv.1.bias"; }

const float *W(omnitok_engine *e, const std::string &k) {
    auto it = e->w.find(k);
    return it == e->w.end() ? nullptr : static_cast<const float *>(it->second.p);
}


__global__ void window_bias_dense_kernel(const float *__restrict__ table, const int64_t *__restrict__ index,
                                         int heads, int ntok, float *__restrict__ dense) {
    // dense[h][kv][q] = table[index[q][kv]][h]   (reference attention.py:277-281)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= heads * ntok * ntok) return;
    const int q = idx % ntok, kv = (idx / ntok) % ntok, h = idx / (ntok * ntok);
    dense[idx] = table[index[q * ntok + kv] * heads + h];
}

__global__ void pad_cols_kernel(const float *__restrict__ src, int rows, int cols, int cols_pad,
                                float *__restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * cols_pad) return;
    const int r = (int)(idx / cols_pad), c = (int)(idx % cols_pad);
    dst[idx] = c < cols ? src[(int64_t)r * cols + c] : 0.0f;
}


static int weight_range(omnitok_engine *e, const float *w, int64_t ld, int rows, int K, float *norm, float *amax,
                        hipStream_t stream) {
    if (!e->range_scratch) OT_HIP(hipMalloc(reinterpret_cast<void **>(&e->range_scratch), 2 * sizeof(float)));
    OT_HIP(hipMemsetAsync(e->range_scratch, 0, 2 * sizeof(float), stream));
    if (int rc = omnitok_weight_range(w, ld, rows, K, e->range_scratch, stream)) return rc;
    float h[2];
    OT_HIP(hipMemcpyAsync(h, e->range_scratch, sizeof(h), hipMemcpyDeviceToHost, stream));
    OT_HIP(hipStreamSynchronize(stream));
    if (norm) *norm = h[0];
    if (amax) *amax = h[1];
    return OMNITOK_OK;
}

// >= max |LayerNorm(x)_k| = |xhat_k gamma_k + beta_k| with |xhat_k| <= sqrt(dim); *l2 (optional) >= ||LN(x)||_2
static int ln_range(omnitok_engine *e, const float *gamma, const float *beta, int dim, float *amax, float *l2,
                    hipStream_t stream) {
    float gn = 0, gm = 0, bn = 0, bm = 0;
    if (int rc = weight_range(e, gamma, dim, 1, dim, &gn, &gm, stream)) return rc;
    if (beta)
        if (int rc = weight_range(e, beta, dim, 1, dim, &bn, &bm, stream)) return rc;
    const float sq = sqrtf((float)dim);
    if (amax) *amax = 1.01f * (sq * gm + bm);
    if (l2) *l2 = 1.01f * (sq * gm + bn);  // ||xhat o gamma|| <= max|gamma| ||xhat|| = max|gamma| sqrt(dim)
    return OMNITOK_OK;
}

// pack a GEMM weight for gemm_h2 and register it under its fp32 pointer
static int pack_h2(omnitok_engine *e, const float *w, int64_t ld, int N, int K, hipStream_t stream) {
    if (!w || N % 32 || K % 32 || ld != K) return OMNITOK_OK;  // shapes the h2 kernel does not take: x3 / fp32 path
    float *pl, *sc;
    if (int rc = alloc_f(e, &pl, (int64_t)((N + 63) / 64 * 64) * K)) return rc;  // 2 planes x 2 B, 64-row blocks
    if (int rc = alloc_f(e, &sc, N)) return rc;
    if (int rc = omnitok_h2_pack_weight(w, ld, N, K, pl, sc, stream)) return rc;
    H2W h;
    h.pl = pl;
    h.sc = sc;
    e->h2w[w] = h;
    return OMNITOK_OK;
}

// pack a GEMM weight for gemm_pl (the activation operand arrives as planes written by its producer)
static int pack_pl(omnitok_engine *e, const float *w, int64_t ld, int N, int K, hipStream_t stream) {
    if (!w || N % 32 || K % 32 || ld != K) return OMNITOK_OK;
    const int n_pad = (N + 255) / 256 * 256;
    float *pl, *sc;
    if (int rc = alloc_f(e, &pl, (int64_t)n_pad * K)) return rc;
    if (int rc = alloc_f(e, &sc, N)) return rc;
    if (int rc = omnitok_pl_pack_weight(w, ld, N, K, n_pad, pl, sc, stream)) return rc;
    H2W h;
    h.pl = pl;
    h.sc = sc;
    e->plw[w] = h;
    return OMNITOK_OK;
}

// LayerNorm folded into a Linear whose operand is the centred row x - mean: rows [0, rows_fold) of w are multiplied by
// gamma and get b[n] = sum_k w[n][k] beta[k]; the other rows are copied and get u[n] = sum_k w[n][k] (fp64 accumulation).
// One wave per row.
__global__ __launch_bounds__(256) void fold_ln_weight_kernel(const float *__restrict__ w, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, int rows_fold, int rows, int K,
                                                             float *__restrict__ wout, float *__restrict__ b,
                                                             float *__restrict__ u) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= rows) return;
    const bool f = n < rows_fold;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float v = w[(int64_t)n * K + k];
        wout[(int64_t)n * K + k] = f ? v * gamma[k] : v;
        acc += f ? (beta ? (double)v * (double)beta[k] : 0.0) : (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        b[n] = f ? (float)acc : 0.0f;
        u[n] = f ? 0.0f : (float)acc;
    }
}

static int fold_ln_weight(omnitok_engine *e, const float *w, const float *gamma, const float *beta, int rows_fold, int rows, int K,
                          const float **wout, const float **b, const float **u, hipStream_t stream) {
    float *wo, *bb, *uu;
    if (int rc = alloc_f(e, &wo, (int64_t)rows * K)) return rc;
    if (int rc = alloc_f(e, &bb, rows)) return rc;
    if (int rc = alloc_f(e, &uu, rows)) return rc;
    hipLaunchKernelGGL(fold_ln_weight_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, w, gamma, beta, rows_fold, rows, K, wo, bb,
                       uu);
    OT_LAUNCH_CHECK("fold_ln_weight");
    if (int rc = pack_pl(e, wo, K, rows, K, stream)) return rc;
    *wout = wo;
    *b = bb;
    *u = uu;
    return OMNITOK_OK;
}

// rows of the folded [to_q o gamma | to_k | to_v] weight regrouped per head for the fused temporal stage: out row
// h * 128 + j = q row h * 64 + j (j < 64) or k row D + h * 64 + (j - 64); fb / fu likewise (fold_b is 0 on k rows, fold_u on q rows)
__global__ __launch_bounds__(256) void regroup_qk_kernel(const float *__restrict__ w, const float *__restrict__ fb,
                                                         const float *__restrict__ fu, int D, int K, float *__restrict__ wout,
                                                         float *__restrict__ fbo, float *__restrict__ fuo) {
    const int n = blockIdx.x;  // output row, 0 .. 2D - 1
    const int h = n >> 7, j = n & 127;
    const int src = j < 64 ? h * 64 + j : D + h * 64 + (j - 64);
    for (int k = threadIdx.x; k < K; k += 256) wout[(int64_t)n * K + k] = w[(int64_t)src * K + k];
    if (threadIdx.x == 0) {
        fbo[n] = fb[src];
        fuo[n] = fu[src];
    }
}

// wvo[o][i] = sum_m wo[o][m] wv[m][i]  (both [D, D] row-major), fp64 accumulation.  One thread per output element.
__global__ __launch_bounds__(256) void compose_vo_kernel(const float *__restrict__ wo, const float *__restrict__ wv, int D,
                                                         float *__restrict__ wvo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= D * D) return;
    const int o = idx / D, i = idx % D;
    double acc = 0.0;
    for (int m = 0; m < D; ++m) acc += (double)wo[(int64_t)o * D + m] * (double)wv[(int64_t)m * D + i];
    wvo[idx] = (float)acc;
}

int build_transformer(omnitok_engine *e, TransformerW &tw, const std::string &prefix, const std::string &block,
                             bool spatial, hipStream_t stream) {
    const omnitok_config &c = e->cfg;
    tw.layers.clear();
    for (size_t i = 0; i < block.size(); ++i) {
        const std::string p = prefix + ".layers." + std::to_string(i);
        Layer L;
        L.kind = block[i];
        if (block[i] == 't') {
            float *w27;
            if (int rc = alloc_f(e, &w27, 27 * (int64_t)c.dim)) return rc;
            if (int rc = omnitok_pack_peg_weight(W(e, p + ".0.dsconv.weight"), c.dim, w27, stream)) return rc;
            L.t.peg_w27 = w27;
            L.t.peg_b = W(e, p + ".0.dsconv.bias");
            L.t.ng = W(e, p + ".1.norm.gamma");
            L.t.nb = W(e, p + ".1.norm.beta");
            L.t.wq = W(e, p + ".1.to_q.weight");
            L.t.wkv = W(e, p + ".1.to_kv.weight");
            L.t.wo = W(e, p + ".1.to_out.weight");
            {
                float *m;
                if (int rc = alloc_f(e, &m, 3 * (int64_t)c.dim * c.dim)) return rc;
                OT_HIP(hipMemcpyAsync(m, L.t.wq, (size_t)c.dim * c.dim * 4, hipMemcpyDeviceToDevice, stream));
                OT_HIP(hipMemcpyAsync(m + (int64_t)c.dim * c.dim, L.t.wkv, (size_t)2 * c.dim * c.dim * 4,
                                      hipMemcpyDeviceToDevice, stream));
                L.t.wqkv = m;
                if (int rc = pack_h2(e, m, c.dim, 3 * c.dim, c.dim, stream)) return rc;
                auto it = e->h2w.find(m);
                if (it != e->h2w.end()) {  // the V rows alone (images: temporal attention over one token)
                    H2W v;
                    v.pl = static_cast<const char *>(it->second.pl) + (int64_t)2 * c.dim * c.dim * 4;
                    v.sc = it->second.sc + 2 * c.dim;
                    e->h2w[L.t.wkv + (int64_t)c.dim * c.dim] = v;
                }
            }
            if (int rc = pack_h2(e, L.t.wo, c.dim, c.dim, c.dim, stream)) return rc;
            if (int rc = pack_pl(e, L.t.wo, c.dim, c.dim, c.dim, stream)) return rc;
            if (c.dim % 256 == 0)
                if (int rc = fold_ln_weight(e, L.t.wqkv, L.t.ng, L.t.nb, c.dim, 3 * c.dim, c.dim, &L.t.wqkv_fold, &L.t.fold_b,
                                            &L.t.fold_u, stream))
                    return rc;
            if (!spatial && L.t.wqkv_fold && c.dim == c.heads * 64 && c.heads % 2 == 0) {
                float *wq, *fb, *fu;
                if (int rc = alloc_f(e, &wq, 2 * (int64_t)c.dim * c.dim)) return rc;
                if (int rc = alloc_f(e, &fb, 2 * c.dim)) return rc;
                if (int rc = alloc_f(e, &fu, 2 * c.dim)) return rc;
                hipLaunchKernelGGL(regroup_qk_kernel, dim3(2 * c.dim), dim3(256), 0, stream, L.t.wqkv_fold, L.t.fold_b, L.t.fold_u,
                                   c.dim, c.dim, wq, fb, fu);
                OT_LAUNCH_CHECK("regroup_qk");
                if (int rc = pack_pl(e, wq, c.dim, 2 * c.dim, c.dim, stream)) return rc;
                L.t.wqk_t = wq;
                L.t.qk_t_fb = fb;
                L.t.qk_t_fu = fu;
            }
            if (int rc = ln_range(e, L.t.ng, L.t.nb, c.dim, &L.t.ln_bound, nullptr, stream)) return rc;
            if (int rc = weight_range(e, L.t.wkv + (int64_t)c.dim * c.dim, c.dim, c.dim, c.dim, &L.t.vnorm, nullptr, stream))
                return rc;
            if (!spatial && c.dim == 512) {  // images run the temporal blocks on one token: Wo . Wv as one packed weight
                float *wvo;
                if (int rc = alloc_f(e, &wvo, (int64_t)c.dim * c.dim)) return rc;
                hipLaunchKernelGGL(compose_vo_kernel, dim3((c.dim * c.dim + 255) / 256), dim3(256), 0, stream, L.t.wo,
                                   L.t.wkv + (int64_t)c.dim * c.dim, c.dim, wvo);
                OT_LAUNCH_CHECK("compose_vo");
                if (int rc = pack_pl(e, wvo, c.dim, c.dim, c.dim, stream)) return rc;
                L.t.wvo = wvo;
            }
            L.t.q_scale = W(e, p + ".1.q_scale");
            L.t.k_scale = W(e, p + ".1.k_scale");
            if (int rc = weight_range(e, L.t.q_scale, c.dim_head, 1, c.dim_head, nullptr, &L.t.q_amax, stream)) return rc;
            if (int rc = weight_range(e, L.t.k_scale, c.dim_head, 1, c.dim_head, nullptr, &L.t.k_amax, stream)) return rc;
            L.t.bias_prefix = (spatial && c.legacy_attention && !c.spatial_rope) ? p + ".1.spatial_rel_pos_bias" : "";
        } else if (block[i] == 'l') {
            L.pool_w = W(e, p + ".1.pool.weight");
            L.pool_b = W(e, p + ".1.pool.bias");
        } else if (block[i] == 'r') {
            L.pool_w = W(e, p + ".1.up.2.weight");
            L.pool_b = W(e, p + ".1.up.2.bias");
        } else if (block[i] == 'a' || block[i] == 'm' || block[i] == 'n') {
        } else {
            const int ntok = c.window_size * c.window_size;
            float *dense;
            if (int rc = alloc_f(e, &dense, (int64_t)c.heads * ntok * ntok)) return rc;
            const int total = c.heads * ntok * ntok;
            hipLaunchKernelGGL(window_bias_dense_kernel, dim3((total + 255) / 256), dim3(256), 0, stream,
                               W(e, p + ".1.relative_position_bias_table"),
                               static_cast<const int64_t *>(e->w[p + ".1.relative_position_index"].p), c.heads, ntok,
                               dense);
            OT_LAUNCH_CHECK("window_bias_dense");
            L.w.ng = W(e, p + ".1.norm.gamma");
            L.w.nb = W(e, p + ".1.norm.beta");
            L.w.wqkv = W(e, p + ".1.qkv.weight");
            L.w.wproj = W(e, p + ".1.proj.weight");
            L.w.bproj = W(e, p + ".1.proj.bias");
            L.w.bias_dense = dense;
            if (int rc = pack_h2(e, L.w.wqkv, c.dim, 3 * c.dim, c.dim, stream)) return rc;
            if (int rc = pack_h2(e, L.w.wproj, c.dim, c.dim, c.dim, stream)) return rc;
            if (int rc = pack_pl(e, L.w.wproj, c.dim, c.dim, c.dim, stream)) return rc;
            if (c.dim % 256 == 0)
                if (int rc = fold_ln_weight(e, L.w.wqkv, L.w.ng, L.w.nb, 3 * c.dim, 3 * c.dim, c.dim, &L.w.wqkv_fold, &L.w.fold_b,
                                            &L.w.fold_u, stream))
                    return rc;
            {
                float l2 = 0, vn = 0;
                if (int rc = ln_range(e, L.w.ng, L.w.nb, c.dim, &L.w.ln_bound, &l2, stream)) return rc;
                if (int rc = weight_range(e, L.w.wqkv + (int64_t)2 * c.dim * c.dim, c.dim, c.dim, c.dim, &vn, nullptr,
                                          stream))
                    return rc;
                L.w.ao_bound = 1.01f * l2 * vn;  // |softmax-weighted V| <= max |V_j| <= ||LN(x)|| ||Wv_j||
                float qn = 0, kn = 0;
                if (int rc = weight_range(e, L.w.wqkv, c.dim, c.dim, c.dim, &qn, nullptr, stream)) return rc;
                if (int rc = weight_range(e, L.w.wqkv + (int64_t)c.dim * c.dim, c.dim, c.dim, c.dim, &kn, nullptr, stream)) return rc;
                L.w.q_bound = 1.01f * l2 * qn / sqrtf((float)c.dim_head);  // q * head_dim^-0.5 (attention.py:227, 274)
                L.w.k_bound = 1.01f * l2 * kn;
            }
        }
        float *w1p, *w2p;
        if (int rc = alloc_f(e, &w1p, 2 * (int64_t)e->inner_pad * c.dim)) return rc;
        if (int rc = omnitok_pack_geglu_weight(W(e, p + ".3.1.weight"), c.ff_inner, c.dim, e->inner_pad, w1p, stream))
            return rc;
        if (int rc = alloc_f(e, &w2p, (int64_t)c.dim * e->inner_pad)) return rc;
        {
            const int64_t total = (int64_t)c.dim * e->inner_pad;
            hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               W(e, p + ".3.4.weight"), c.dim, c.ff_inner, e->inner_pad, w2p);
            OT_LAUNCH_CHECK("pad_cols");
        }
        L.ff.lw = W(e, p + ".3.0.weight");
        L.ff.lb = W(e, p + ".3.0.bias");
        L.ff.w1p = w1p;
        L.ff.w2p = w2p;
        if (int rc = pack_h2(e, w1p, c.dim, 2 * e->inner_pad, c.dim, stream)) return rc;
        if (int rc = pack_h2(e, w2p, e->inner_pad, c.dim, e->inner_pad, stream)) return rc;
        if (int rc = pack_pl(e, w1p, c.dim, 2 * e->inner_pad, c.dim, stream)) return rc;
        if (int rc = pack_pl(e, w2p, e->inner_pad, c.dim, e->inner_pad, stream)) return rc;
        {
            float l2 = 0, wn = 0;
            if (int rc = ln_range(e, L.ff.lw, L.ff.lb, c.dim, &L.ff.ln_bound, &l2, stream)) return rc;
            if (int rc = weight_range(e, w1p, c.dim, 2 * e->inner_pad, c.dim, &wn, nullptr, stream)) return rc;
            // |gelu(g) v| <= |g| |v| <= (||LN(x)|| max_j ||W1_j||)^2
            L.ff.h_bound = 1.01f * (l2 * wn) * (l2 * wn);
        }
        tw.layers.push_back(L);
    }
    tw.og = W(e, prefix + ".norm_out.gamma");
    tw.ob = W(e, prefix + ".norm_out.beta");
    if (int rc = ln_range(e, tw.og, tw.ob, c.dim, &tw.out_bound, nullptr, stream)) return rc;
    return OMNITOK_OK;
}

// eval-mode (Sync)BatchNorm folded into the preceding convolution (ATen's CPU batch_norm applies
// y = x * alpha + beta with alpha = w / sqrt(var + eps), beta = b - mean * alpha):
//   encoder  Conv3d weight [D, K] (K = c pt p1 p2, the patchify feature order): row d scaled by alpha[d]
__global__ void fold_bn_rows_kernel(const float *__restrict__ w, const float *__restrict__ cb,
                                    const float *__restrict__ bw, const float *__restrict__ bb,
                                    const float *__restrict__ mean, const float *__restrict__ var, float eps, int rows,
                                    int K, int ld, float *__restrict__ wo, float *__restrict__ bo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * ld) return;
    const int r = (int)(idx / ld), k = (int)(idx % ld);
    const float alpha = bw[r] / sqrtf(var[r] + eps);
    wo[idx] = k < K ? w[(int64_t)r * K + k] * alpha : 0.0f;
    if (k == 0) bo[r] = cb[r] * alpha + (bb[r] - mean[r] * alpha);
}
//   decoder  ConvTranspose3d weight [D, N] (N = c pt p1 p2) -> Linear layout [N, D], column n scaled by
//   the alpha of its output channel c = n / per_c
__global__ void fold_bn_transpose_kernel(const float *__restrict__ w, const float *__restrict__ cb,
                                         const float *__restrict__ bw, const float *__restrict__ bb,
                                         const float *__restrict__ mean, const float *__restrict__ var, float eps,
                                         int D, int N, int per_c, float *__restrict__ wo, float *__restrict__ bo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * D) return;
    const int n = (int)(idx / D), k = (int)(idx % D);
    const int ch = n / per_c;
    const float alpha = bw[ch] / sqrtf(var[ch] + eps);
    wo[idx] = w[(int64_t)k * N + n] * alpha;
    if (k == 0) bo[n] = cb[ch] * alpha + (bb[ch] - mean[ch] * alpha);
}

int build_patch_operands(omnitok_engine *e, hipStream_t stream) {
    const omnitok_config &c = e->cfg;
    const Geo g = geometry(c);
    const int D = c.dim, C = c.image_channels;
    const char *names[2] = {"encoder.to_patch_emb_first_frame", "encoder.to_patch_emb"};
    const char *pnames[2] = {"decoder.to_pixels_first_frame", "decoder.to_pixels"};
    for (int i = 0; i < 2; ++i) {
        const std::string p = names[i];
        const int K = C * g.p_enc * g.p_enc * (i ? g.pt_enc : 1);
        const int ld = ((K + 31) / 32) * 32;  // the GEMM's K % 32 rule; patchify zero-fills the pad
        e->pe_k[i] = K;
        e->pe_ld[i] = ld;
        if (c.patch_embed_cnn) {
            float *wo, *bo;
            if (int rc = alloc_f(e, &wo, (int64_t)D * ld)) return rc;
            if (int rc = alloc_f(e, &bo, D)) return rc;
            const int64_t total = (int64_t)D * ld;
            hipLaunchKernelGGL(fold_bn_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               W(e, p + ".0.weight"), W(e, p + ".0.bias"), W(e, p + ".1.weight"), W(e, p + ".1.bias"),
                               W(e, p + ".1.running_mean"), W(e, p + ".1.running_var"), 1e-5f, D, K, ld, wo, bo);
            OT_LAUNCH_CHECK("fold_bn_rows");
            e->pe_w[i] = wo;
            e->pe_b[i] = bo;
        } else if (ld != K) {
            float *wo;
            if (int rc = alloc_f(e, &wo, (int64_t)D * ld)) return rc;
            const int64_t total = (int64_t)D * ld;
            hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               W(e, p + ".2.weight"), D, K, ld, wo);
            OT_LAUNCH_CHECK("pad_cols");
            e->pe_w[i] = wo;
            e->pe_b[i] = W(e, p + ".2.bias");
        } else {
            e->pe_w[i] = W(e, p + ".2.weight");
            e->pe_b[i] = W(e, p + ".2.bias");
        }
        const std::string q = pnames[i];
        if (c.patch_embed_cnn) {
            const int per_c = g.p_dec * g.p_dec * (i ? g.pt_dec : 1), N = C * per_c;
            float *wo, *bo;
            if (int rc = alloc_f(e, &wo, (int64_t)N * D)) return rc;
            if (int rc = alloc_f(e, &bo, N)) return rc;
            const int64_t total = (int64_t)N * D;
            hipLaunchKernelGGL(fold_bn_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               W(e, q + ".1.weight"), W(e, q + ".1.bias"), W(e, q + ".2.weight"), W(e, q + ".2.bias"),
                               W(e, q + ".2.running_mean"), W(e, q + ".2.running_var"), 1e-5f, D, N, per_c, wo, bo);
            OT_LAUNCH_CHECK("fold_bn_transpose");
            e->px_w[i] = wo;
            e->px_b[i] = bo;
        } else {
            e->px_w[i] = W(e, q + ".0.weight");
            e->px_b[i] = W(e, q + ".0.bias");
        }
        e->pe_bound[i] = 0.0f;
        if (!c.patch_embed_cnn) {  // the patch rows are LayerNorm outputs: a static range exists
            if (int rc = ln_range(e, W(e, p + ".1.weight"), W(e, p + ".1.bias"), K, &e->pe_bound[i], nullptr, stream))
                return rc;
            if (int rc = pack_h2(e, e->pe_w[i], ld, D, ld, stream)) return rc;
            const int Kp = C * g.p_dec * g.p_dec * (i ? g.pt_dec : 1);
            if (int rc = pack_h2(e, e->px_w[i], D, Kp, D, stream)) return rc;
            // plane data flow (gemm_pl.h): the pixel GEMM's operand comes from the norm_out pass as planes, the un-patchify
            // store is its epilogue
            if (Kp % 32 == 0)
                if (int rc = pack_pl(e, e->px_w[i], D, Kp, D, stream)) return rc;
        }
    }
    return OMNITOK_OK;
}

int get_rope(omnitok_engine *e, int N, const float **cosp, const float **sinp, hipStream_t stream) {
    auto it = e->rope.find(N);
    if (it == e->rope.end()) {
        const int half = e->cfg.dim_head / 2;
        std::vector<float> hc((size_t)N * half), hs((size_t)N * half);
        if (int rc = omnitok_rope_table(N, e->cfg.dim_head, 10000.0f, hc.data(), hs.data())) return rc;
        float *dc, *ds;
        if (int rc = alloc_f(e, &dc, (int64_t)N * half)) return rc;
        if (int rc = alloc_f(e, &ds, (int64_t)N * half)) return rc;
        // synchronous copies: the host vectors die at scope exit
        OT_HIP(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        OT_HIP(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
        it = e->rope.emplace(N, std::make_pair(dc, ds)).first;
    }
    *cosp = it->second.first;
    *sinp = it->second.second;
    return OMNITOK_OK;
}

// ContinuousPositionBias evaluated once per distinct (dy,dx) offset (reference attention.py:535-583;
// the bias of a token pair depends only on its offset, :567-574): table[(2gh-1)*(2gw-1), heads].
int get_bias_table(omnitok_engine *e, const std::string &prefix, int gh, int gw, const float **out,
                          hipStream_t stream) {
    const std::string key = prefix + "|" + std::to_string(gh) + "|" + std::to_string(gw);
    auto it = e->bias_tables.find(key);
    if (it != e->bias_tables.end()) {
        *out = it->second;
        return OMNITOK_OK;
    }
    const omnitok_config &c = e->cfg;
    const int P = (2 * gh - 1) * (2 * gw - 1);
    // input features sign(d)*log(1+|d|), K padded 2 -> 32 for the GEMM's K % 32 rule
    std::vector<float> feat((size_t)P * 32, 0.0f);
    for (int dy = -(gh - 1); dy <= gh - 1; ++dy)
        for (int dx = -(gw - 1); dx <= gw - 1; ++dx) {
            const int i = (dy + gh - 1) * (2 * gw - 1) + (dx + gw - 1);
            auto f = [](int d) {
                const float s = d > 0 ? 1.0f : (d < 0 ? -1.0f : 0.0f);
                return s * logf((float)std::abs(d) + 1.0f);
            };
            feat[(size_t)i * 32 + 0] = f(dy);
            feat[(size_t)i * 32 + 1] = f(dx);
        }
    float *dfeat, *w0p, *h0, *h1, *tab;
    if (int rc = alloc_f(e, &dfeat, (int64_t)P * 32)) return rc;
    if (int rc = alloc_f(e, &w0p, (int64_t)c.dim * 32)) return rc;
    if (int rc = alloc_f(e, &h0, (int64_t)P * c.dim)) return rc;
    if (int rc = alloc_f(e, &h1, (int64_t)P * c.dim)) return rc;
    if (int rc = alloc_f(e, &tab, (int64_t)P * c.heads)) return rc;
    OT_HIP(hipMemcpy(dfeat, feat.data(), feat.size() * 4, hipMemcpyHostToDevice));
    {
        const int64_t total = (int64_t)c.dim * 32;
        hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           W(e, prefix + ".net.0.0.weight"), c.dim, 2, 32, w0p);
        OT_LAUNCH_CHECK("pad_cols");
    }
    const int BL = OMNITOK_GEMM_BIAS | OMNITOK_GEMM_LEAKY;
    if (int rc = omnitok_gemm(dfeat, 32, w0p, 32, W(e, prefix + ".net.0.0.bias"), nullptr, 0, h0, c.dim, P, c.dim, 32,
                              BL, 0, 0, 0, stream))
        return rc;
    if (int rc = omnitok_gemm(h0, c.dim, W(e, prefix + ".net.1.0.weight"), c.dim, W(e, prefix + ".net.1.0.bias"),
                              nullptr, 0, h1, c.dim, P, c.dim, c.dim, BL, 0, 0, 0, stream))
        return rc;
    if (int rc = omnitok_gemm(h1, c.dim, W(e, prefix + ".net.2.weight"), c.dim, W(e, prefix + ".net.2.bias"), nullptr,
                              0, tab, c.heads, P, c.heads, c.dim, OMNITOK_GEMM_BIAS, 0, 0, 0, stream))
        return rc;
    e->bias_tables[key] = tab;
    *out = tab;
    return OMNITOK_OK;
}

// GEMM dispatch of the engine.  "gemm_mode"
//   2 (default): 2-way fp16 split, three fp16-MFMA products (gemm_h2.hip) wherever a rigorous range of the A
//      operand is known (LayerNorm outputs: static; raw activations / attention outputs: from the row
//      statistics pass; GEGLU hidden: from the weights), bf16x3 elsewhere;
//   1: fp32 operands split in-kernel into three bf16 planes, six bf16-MFMA products (gemm_x3.hip);
//   0: the fp32-input MFMA kernels of gemm.hip (bitwise an fmaf chain).
// All three have fp32-class error (tests/test_gpu_ops.py).  The mode is process-wide and the per-element

}  // namespace omnitok

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------

extern "C" int omnitok_rope_table(int n_tokens, int dim_head, float theta, float *host_cos, float *host_sin) {
    OT_CHECK_ARG(n_tokens > 0 && dim_head % 4 == 0 && host_cos && host_sin, "rope_table: bad arguments");
    // reference attention.py:28-43: H = int(sqrt(N)); x = pos % H, y = pos // H;
    // freqs_i = 1 / theta^(4i/dim) (fp32); angles (x*f_i, y*f_i) interleaved; cos/sin via polar
    const int H = (int)std::sqrt((double)n_tokens);
    const int nf = dim_head / 4, half = dim_head / 2;
    std::vector<float> freqs(nf);
    for (int i = 0; i < nf; ++i) freqs[i] = 1.0f / powf(theta, (float)(4 * i) / (float)dim_head);
    for (int n = 0; n < n_tokens; ++n) {
        const float xp = (float)(n % H), yp = (float)(n / H);
        for (int i = 0; i < nf; ++i) {
            const float ax = xp * freqs[i], ay = yp * freqs[i];  // fp32 products like torch.outer
            host_cos[(size_t)n * half + 2 * i] = (float)std::cos((double)ax);
            host_sin[(size_t)n * half + 2 * i] = (float)std::sin((double)ax);
            host_cos[(size_t)n * half + 2 * i + 1] = (float)std::cos((double)ay);
            host_sin[(size_t)n * half + 2 * i + 1] = (float)std::sin((double)ay);
        }
    }
    return OMNITOK_OK;
}


extern "C" int omnitok_engine_create(const omnitok_config *cfg, omnitok_engine **out) {
    OT_CHECK_ARG(cfg && out, "engine_create: null pointer");
    const omnitok_config &c = *cfg;
    if (c.dim_head != 64 || c.dim != c.heads * c.dim_head || c.dim % 128 != 0) {
        set_error("engine_create: kernels are built for dim_head == 64, dim == heads*64, dim %% 128 == 0");
        return OMNITOK_ERR_UNSUPPORTED;
    }
    if (c.codebook_dim != 8 || c.n_codes % 32 != 0 || c.n_codes > 32768) {
        set_error("engine_create: need codebook_dim == 8 and n_codes %% 32 == 0 (<= 32768)");
        return OMNITOK_ERR_UNSUPPORTED;
    }
    if (c.window_size != 8 && (strchr(c.enc_block, 'w') || strchr(c.dec_block, 'w'))) {
        set_error("engine_create: window attention is built for twod_window_size == 8 (got %d)", c.window_size);
        return OMNITOK_ERR_UNSUPPORTED;
    }
    const Geo g = geometry(c);
    if (g.p_enc % 4 != 0 || g.p_dec % 4 != 0 || g.pt_enc < 1 ||
        (c.defer_spatial_pool && c.patch_size % 2) || (c.defer_temporal_pool && c.temporal_patch_size % 2)) {
        set_error("engine_create: patch_size %d / temporal_patch_size %d unsupported (encoder patch %d, decoder patch "
                  "%d must be multiples of 4)", c.patch_size, c.temporal_patch_size, g.p_enc, g.p_dec);
        return OMNITOK_ERR_UNSUPPORTED;
    }
    if ((int64_t)c.image_channels * g.p_enc * g.p_enc * g.pt_enc > 1024) {
        set_error("engine_create: patch feature dim > 1024 unsupported");
        return OMNITOK_ERR_UNSUPPORTED;
    }
    const std::string eb(c.enc_block), db(c.dec_block);
    for (char ch : eb)
        if (!strchr("twamlnr", ch)) {
            set_error("engine_create: enc_block type '%c' unknown (reference attention.py:614-649)", ch);
            return OMNITOK_ERR_UNSUPPORTED;
        }
    for (char ch : db)
        if (ch != 't' && ch != 'w') {
            set_error("engine_create: dec_block type '%c' not built (in the decoder 'n'/'r' Up blocks make the "
                      "reference raise at omnitokenizer.py:1078; pooling blocks are encoder-side)", ch);
            return OMNITOK_ERR_UNSUPPORTED;
        }
    omnitok_engine *e = new omnitok_engine();
    e->cfg = c;
    e->inner_pad = ((c.ff_inner + 63) / 64) * 64;
    if (e->inner_pad % 32 != 0) e->inner_pad = ((e->inner_pad + 31) / 32) * 32;
    const int64_t d = c.dim, C = c.image_channels;
    const char *names[2] = {"to_patch_emb_first_frame", "to_patch_emb"};
    const int64_t pts[2] = {1, g.pt_enc};
    for (int i = 0; i < 2; ++i) {
        const std::string p = std::string("encoder.") + names[i];
        const int64_t k = C * g.p_enc * g.p_enc * pts[i];
        if (c.patch_embed_cnn) {  // Conv3d + SyncBatchNorm, reference omnitokenizer.py:823-836
            e->spec[p + ".0.weight"] = {d, C, pts[i], g.p_enc, g.p_enc};
            e->spec[p + ".0.bias"] = {d};
            for (const char *q : {".1.weight", ".1.bias", ".1.running_mean", ".1.running_var"}) e->spec[p + q] = {d};
            continue;
        }
        e->spec[p + ".1.weight"] = {k};
        e->spec[p + ".1.bias"] = {k};
        e->spec[p + ".2.weight"] = {d, k};
        e->spec[p + ".2.bias"] = {d};
        e->spec[p + ".3.weight"] = {d};
        e->spec[p + ".3.bias"] = {d};
    }
    const std::string tb(c.temporal_depth, 't');
    add_transformer_spec(e, "encoder.enc_spatial_transformer", eb, !c.spatial_rope);
    add_transformer_spec(e, "encoder.enc_temporal_transformer", tb, false);
    add_transformer_spec(e, "decoder.dec_spatial_transformer", db, !c.spatial_rope);
    add_transformer_spec(e, "decoder.dec_temporal_transformer", tb, false);
    const char *pnames[2] = {"to_pixels_first_frame", "to_pixels"};
    const int64_t ptd[2] = {1, g.pt_dec};
    for (int i = 0; i < 2; ++i) {
        const std::string p = std::string("decoder.") + pnames[i];
        if (c.patch_embed_cnn) {  // ConvTranspose3d + SyncBatchNorm(3), reference omnitokenizer.py:1019-1031
            e->spec[p + ".1.weight"] = {d, C, ptd[i], g.p_dec, g.p_dec};
            e->spec[p + ".1.bias"] = {C};
            for (const char *q : {".2.weight", ".2.bias", ".2.running_mean", ".2.running_var"}) e->spec[p + q] = {C};
        } else {
            e->spec[p + ".0.weight"] = {C * g.p_dec * g.p_dec * ptd[i], d};
            e->spec[p + ".0.bias"] = {C * g.p_dec * g.p_dec * ptd[i]};
        }
    }
    // --use_vae: no quantiser on the path, pre_vq emits mean | logvar (reference omnitokenizer.py:143-154)
    const int64_t pre_out = c.use_vae ? 2 * c.codebook_dim : c.codebook_dim;
    if (c.external_codebook) {  // VectorQuantize + CosineSimCodebook (vector_quantize_pytorch.py:514, 690)
        if (c.use_vae) {
            set_error("engine_create: use_vae with the external codebook is not a valid reference configuration");
            delete e;
            return OMNITOK_ERR_UNSUPPORTED;
        }
        e->spec[k_embed(c)] = {1, c.n_codes, c.codebook_dim};
    } else if (!c.use_vae) {
        e->spec[k_embed(c)] = {c.n_codes, c.codebook_dim};
    }
    e->spec[k_pre_w(c)] = {pre_out, d};
    e->spec[k_pre_b(c)] = {pre_out};
    e->spec[k_post_w(c)] = {d, c.codebook_dim};
    e->spec[k_post_b(c)] = {d};
    // drop keys the inference path never reads
    for (auto it = e->spec.begin(); it != e->spec.end();)
        it = key_unused(e, it->first) ? e->spec.erase(it) : std::next(it);
    *out = e;
    return OMNITOK_OK;
}

extern "C" void omnitok_engine_destroy(omnitok_engine *e) {
    if (!e) return;
    for (auto &kv : e->w)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (void *p : e->owned) (void)hipFree(p);
    if (!e->ext_ws)
        for (Buf *b : {&e->X, &e->X2, &e->Y, &e->QKV, &e->AO, &e->HD, &e->Z, &e->ST})
            if (b->p) (void)hipFree(b->p);
    if (e->err_flag) (void)hipFree(e->err_flag);
    if (e->bounds) (void)hipFree(e->bounds);
    for (void *p : e->retired) (void)hipFree(p);
    if (e->range_scratch) (void)hipFree(e->range_scratch);
    for (auto &r : e->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    delete e;
}

extern "C" int omnitok_engine_set_weight(omnitok_engine *e, const char *name, const void *dev_ptr,
                                         const int64_t *shape, int ndim, int is_int64, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(e && name && dev_ptr && shape, "set_weight: null pointer");
    auto it = e->spec.find(name);
    if (it == e->spec.end()) return 1;  // off-path / unused key: ignored like strict=False
    std::vector<int64_t> shp(shape, shape + ndim);
    if (shp != it->second) {
        std::string want, got;
        for (auto s : it->second) want += std::to_string(s) + ",";
        for (auto s : shp) got += std::to_string(s) + ",";
        set_error("set_weight: size mismatch for %s: expected [%s] got [%s]", name, want.c_str(), got.c_str());
        return OMNITOK_ERR_INVALID;
    }
    const bool want_i64 = std::string(name).find("relative_position_index") != std::string::npos;
    OT_CHECK_ARG(want_i64 == (is_int64 != 0), "set_weight: dtype mismatch for %s", name);
    DevTensor &t = e->w[name];
    t.shape = shp;
    t.is_int64 = want_i64;
    const size_t bytes = (size_t)t.numel() * (want_i64 ? 8 : 4);
    if (!t.p) OT_HIP(hipMalloc(&t.p, bytes));
    OT_HIP(hipMemcpyAsync(t.p, dev_ptr, bytes, hipMemcpyDeviceToDevice, stream));
    e->finalized = false;
    return OMNITOK_OK;
}

extern "C" int omnitok_engine_missing(omnitok_engine *e, char *buf, int buflen) {
    int n = 0;
    std::string s;
    for (auto &kv : e->spec)
        if (!e->w.count(kv.first)) {
            ++n;
            s += kv.first + "\n";
        }
    if (buf && buflen > 0) {
        strncpy(buf, s.c_str(), buflen - 1);
        buf[buflen - 1] = 0;
    }
    return n;
}

extern "C" int omnitok_engine_finalize(omnitok_engine *e, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(e, "finalize: null engine");
    for (auto &kv : e->spec)
        if (!e->w.count(kv.first)) {
            set_error("finalize: missing weight %s", kv.first.c_str());
            return OMNITOK_ERR_STATE;
        }
    // (re)build derived buffers
    OT_HIP(hipStreamSynchronize(stream));
    for (void *p : e->owned) (void)hipFree(p);
    e->owned.clear();
    e->h2w.clear();
    e->plw.clear();
    e->rope.clear();
    e->bias_tables.clear();
    const omnitok_config &c = e->cfg;
    const std::string tb(c.temporal_depth, 't');
    if (int rc = build_transformer(e, e->enc_s, "encoder.enc_spatial_transformer", c.enc_block, true, stream)) return rc;
    if (int rc = build_transformer(e, e->enc_t, "encoder.enc_temporal_transformer", tb, false, stream)) return rc;
    if (int rc = build_transformer(e, e->dec_s, "decoder.dec_spatial_transformer", c.dec_block, true, stream)) return rc;
    if (int rc = build_transformer(e, e->dec_t, "decoder.dec_temporal_transformer", tb, false, stream)) return rc;
    if (int rc = build_patch_operands(e, stream)) return rc;
    if (!c.use_vae) {
        if (int rc = alloc_f(e, &e->cb_packed, (int64_t)c.n_codes * 8)) return rc;
        if (int rc = alloc_f(e, &e->cb_ee, c.n_codes)) return rc;
        if (int rc = omnitok_vq_prepare(W(e, k_embed(c)), c.n_codes, 8, e->cb_packed, e->cb_ee, stream))
            return rc;
        if (!c.external_codebook) {  // fp16 screening fragments + norm bounds of the screened search (vq.hip)
            if (int rc = alloc_f(e, &e->cb_screen, (int64_t)c.n_codes * 8 + 4)) return rc;
            if (int rc = omnitok_vq_screen_prepare(W(e, k_embed(c)), e->cb_ee, c.n_codes, 8, e->cb_screen, stream)) return rc;
        }
        // F.embedding + post_vq_conv as one table (reference omnitokenizer.py:270, 156-160)
        float *scratch;
        if (int rc = alloc_f(e, &e->dq_table, (int64_t)c.n_codes * c.dim)) return rc;
        if (int rc = alloc_f(e, &scratch, (int64_t)c.n_codes * 2)) return rc;
        if (int rc = omnitok_dequant_table(W(e, k_embed(c)), c.n_codes, 8, W(e, k_post_w(c)), W(e, k_post_b(c)),
                                           e->dq_table, c.dim, reinterpret_cast<int64_t *>(scratch), stream))
            return rc;
    }
    {
        // ALiBi slopes, reference attention.py:506-517 (_get_slopes)
        std::vector<float> sl;
        auto pow2 = [](int n) {
            std::vector<double> v;
            const double start = std::pow(2.0, -std::pow(2.0, -(std::log2((double)n) - 3.0)));
            for (int i = 0; i < n; ++i) v.push_back(start * std::pow(start, i));
            return v;
        };
        const int h = c.heads;
        if ((h & (h - 1)) == 0) {
            for (double v : pow2(h)) sl.push_back((float)v);
        } else {
            int cp = 1;
            while (cp * 2 <= h) cp *= 2;
            for (double v : pow2(cp)) sl.push_back((float)v);
            auto ext = pow2(2 * cp);
            for (int i = 0; (int)sl.size() < h; i += 2) sl.push_back((float)ext[i]);
        }
        if (int rc = alloc_f(e, &e->alibi, h)) return rc;
        OT_HIP(hipMemcpy(e->alibi, sl.data(), h * 4, hipMemcpyHostToDevice));
    }
    if (!e->err_flag) {
        OT_HIP(hipMalloc(reinterpret_cast<void **>(&e->err_flag), sizeof(int)));
        OT_HIP(hipMemset(e->err_flag, 0, sizeof(int)));
    }
    e->finalized = true;
    return OMNITOK_OK;
}

// Shapes of the path (pooling blocks, deferred pools and gen_upscale change them).
