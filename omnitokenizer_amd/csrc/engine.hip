// Engine: VQGAN.encode() / VQGAN.decode() as one native call each.
//
// Holds device copies of the reference state_dict tensors (by reference key name), the repacked
// derivatives the kernels want (GEGLU-interleaved FF weights, [27,D] PEG weights, A-fragment
// codebook, dense window bias, K-padded FF-out weight), a grow-only workspace, and chains the
// per-operator kernels in the reference's order:
//   encode: patch-embed -> spatial Transformer -> temporal Transformer -> pre_vq -> l2norm -> VQ
//           (reference omnitokenizer.py:247-258, 919-947, 881-916)
//   decode: embedding -> post_vq -> temporal Transformer -> spatial Transformer -> to_pixels
//           (reference omnitokenizer.py:268-291, 1101-1118, 1059-1098)
// Tokens are kept physically in the layout the reference has at each point ('(b t)(h w) d' for
// spatial stages, '(b h w) t d' for temporal stages) so that PEG's raw-buffer view
// (attention.py:319) is reproduced by construction.
#include "common.h"

#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <iterator>

namespace omnitok {

struct DevTensor {
    void *p = nullptr;
    std::vector<int64_t> shape;
    bool is_int64 = false;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct Buf {
    float *p = nullptr;
    int64_t cap = 0;  // floats
};

struct LayerT {  // 't' block (+ FF)
    const float *peg_w27, *peg_b;
    const float *ng, *nb;
    const float *wq, *wkv, *wo;
    const float *wqkv;  // [3D, D] = to_q | to_kv rows (one launch: Q from LN(x), K/V from x)
    const float *q_scale, *k_scale;
    float ln_bound = 0.0f;  // >= max |LN(x)|            (ranges for the fp16-split GEMM, gemm_h2.hip)
    float vnorm = 0.0f;     // max_j ||Wv_j||_2: |attention output| <= max_rows ||x||_2 * vnorm
    float q_amax = 0.0f, k_amax = 0.0f;  // max |q_scale|, max |k_scale|: ranges of the l2-normalised q / k (attn_h2.hip)
    std::string bias_prefix;  // spatial_rel_pos_bias prefix ("" if none)
};
struct LayerW {  // 'w' block
    const float *ng, *nb, *wqkv, *wproj, *bproj, *bias_dense;
    float ln_bound = 0.0f, ao_bound = 0.0f;  // >= max |LN(x)|, >= max |window attention output|
};
struct LayerFF {
    const float *lw, *lb, *w1p, *w2p;
    float ln_bound = 0.0f, h_bound = 0.0f;   // >= max |LN(x)|, >= max |GEGLU hidden|
};
struct H2W {  // a weight packed for gemm_h2 (fp16 hi|lo planes + row scales)
    const void *pl = nullptr;
    const float *sc = nullptr;
};
struct ABound {  // upper bound of |A| of a row of clip c = stat * (dev ? dev[2 c] : 1); stat <= 0: unknown
    float stat = 0.0f;
    const float *dev = nullptr;  // per-clip range slots written by omnitok_row_stats
    int64_t rpc = 0;             // rows per clip
};
struct Layer {
    char kind;  // 't', 'w', a pooling block 'a' / 'm' / 'l', or an Up block 'n' / 'r' (encoder only)
    LayerT t;
    LayerW w;
    LayerFF ff;
    const float *pool_w = nullptr, *pool_b = nullptr;  // 'l': Linear(4*dim, dim); 'r': Linear(dim, dim)
};

// patch / grid geometry of a configuration (reference omnitokenizer.py:792-804, 957-959, 985-1003)
struct Geo {
    bool defer_t, defer_s;
    int p_enc, pt_enc, p_dec, pt_dec, n_pool;
};
struct TransformerW {
    std::vector<Layer> layers;
    const float *og, *ob;
    float out_bound = 0.0f;  // >= max |LN_out(x)|
};

struct TimingRec {
    std::string name;
    hipEvent_t a, b;
    double work;
};

}  // namespace omnitok

using namespace omnitok;

struct omnitok_engine {
    omnitok_config cfg;
    std::map<std::string, std::vector<int64_t>> spec;  // required key -> shape
    std::map<std::string, DevTensor> w;
    std::vector<void *> owned;  // derived buffers
    std::vector<void *> retired;  // outgrown blocks a captured graph may still address (freed with the engine)
    // arithmetic / data-flow modes of THIS engine: -1 = follow the process default (omnitok_set_option), else the value
    // given to omnitok_engine_set_option.  Two engines of one process can run different modes.
    int opt_gemm_mode = -1, opt_attn_mode = -1, opt_attn_vpack = -1, opt_gemm_pl = -1;
    bool finalized = false;
    int inner_pad = 0;
    TransformerW enc_s, enc_t, dec_s, dec_t;
    // patch-embed / to_pixels GEMM operands, [0] first frame, [1] the rest (original tensors, or
    // K-padded / BatchNorm-folded / transposed copies made by finalize)
    const float *pe_w[2] = {nullptr, nullptr}, *pe_b[2] = {nullptr, nullptr};
    int pe_k[2] = {0, 0}, pe_ld[2] = {0, 0};
    const float *px_w[2] = {nullptr, nullptr}, *px_b[2] = {nullptr, nullptr};
    // derived
    float *cb_packed = nullptr, *cb_ee = nullptr, *alibi = nullptr;
    float *dq_table = nullptr;  // [n_codes, dim] = E . post_vq^T + b (decode = row gather)
    std::map<int, std::pair<float *, float *>> rope;                    // N -> cos, sin
    std::map<std::string, float *> bias_tables;                          // prefix|gh|gw -> table
    // workspace: grow-only hipMalloc buffers, or slices of a caller-supplied block (omnitok_engine_set_workspace:
    // the Python mirror hands over memory of PyTorch's caching allocator)
    Buf X, X2, Y, QKV, AO, HD, Z, ST;
    float *ext_ws = nullptr;
    int64_t ext_ws_bytes = 0;
    // fp16-split GEMM (gemm_mode 2): packed weights by fp32 weight pointer, device-side range slots
    std::map<const float *, H2W> h2w;
    std::map<const float *, H2W> plw;  // ... packed for the plane x plane GEMM (gemm_pl.h: rows permuted, padded to 256)
    float pe_bound[2] = {0.0f, 0.0f};
    float *bounds = nullptr;  // [launch][clip][2] range slots, zeroed at the start of every encode / decode
    int bound_next = 0, bound_clips = 0, bound_cap = 0;
    float *range_scratch = nullptr;
    int *err_flag = nullptr;
    // timing
    bool timing = false;
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> pool;
};

namespace omnitok {

static Geo geometry(const omnitok_config &c) {
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
static bool walk_enc_grid(const char *block, int *gh, int *gw, int64_t *peak) {
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

static void add_transformer_spec(omnitok_engine *e, const std::string &prefix, const std::string &block,
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
static bool key_unused(const omnitok_engine *e, const std::string &k) {
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

static int alloc_f(omnitok_engine *e, float **out, int64_t n) {
    void *p = nullptr;
    OT_HIP(hipMalloc(&p, (size_t)n * sizeof(float)));
    e->owned.push_back(p);
    *out = static_cast<float *>(p);
    return OMNITOK_OK;
}

static int ensure(Buf &b, int64_t n) {
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
static const char *k_embed(const omnitok_config &c) { return c.external_codebook ? "codebook._codebook.embed" : "codebook.embeddings"; }
static const char *k_pre_w(const omnitok_config &c) { return c.external_codebook ? "codebook.project_in.weight" : "pre_vq_conv.1.weight"; }
static const char *k_pre_b(const omnitok_config &c) { return c.external_codebook ? "codebook.project_in.bias" : "pre_vq_conv.1.bias"; }
static const char *k_post_w(const omnitok_config &c) { return c.external_codebook ? "codebook.project_out.weight" : "post_vq_conv.1.weight"; }
static const char *k_post_b(const omnitok_config &c) { return c.external_codebook ? "codebook.project_out.bias" : "post_vq_conv.1.bias"; }

static const float *W(omnitok_engine *e, const std::string &k) {
    auto it = e->w.find(k);
    return it == e->w.end() ? nullptr : static_cast<const float *>(it->second.p);
}

// ---- timing -------------------------------------------------------------------------------
struct Scope {
    omnitok_engine *e;
    hipStream_t s;
    size_t idx;
    bool on;
    Scope(omnitok_engine *e_, hipStream_t s_, const char *name, double work) : e(e_), s(s_), on(e_->timing) {
        if (!on) return;
        TimingRec r;
        r.name = name;
        r.work = work;
        for (hipEvent_t *ev : {&r.a, &r.b}) {
            if (!e->pool.empty()) {
                *ev = e->pool.back();
                e->pool.pop_back();
            } else {
                (void)hipEventCreate(ev);
            }
        }
        (void)hipEventRecord(r.a, s);
        e->recs.push_back(r);
        idx = e->recs.size() - 1;
    }
    ~Scope() {
        if (on) (void)hipEventRecord(e->recs[idx].b, s);
    }
};

#define OT_RUN(name, work, call)                  \
    do {                                          \
        Scope _sc(e, stream, name, (double)(work)); \
        int _rc = (call);                         \
        if (_rc != OMNITOK_OK) return _rc;        \
    } while (0)

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

constexpr int N_BOUND_LAUNCHES = 64;  // row-statistics launches with ranges per encode / decode

// (max_rows ||w_row||_2, max |w|) of a device matrix; synchronous (finalize only)
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

static int build_transformer(omnitok_engine *e, TransformerW &tw, const std::string &prefix, const std::string &block,
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
            if (int rc = ln_range(e, L.t.ng, L.t.nb, c.dim, &L.t.ln_bound, nullptr, stream)) return rc;
            if (int rc = weight_range(e, L.t.wkv + (int64_t)c.dim * c.dim, c.dim, c.dim, c.dim, &L.t.vnorm, nullptr, stream))
                return rc;
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
            {
                float l2 = 0, vn = 0;
                if (int rc = ln_range(e, L.w.ng, L.w.nb, c.dim, &L.w.ln_bound, &l2, stream)) return rc;
                if (int rc = weight_range(e, L.w.wqkv + (int64_t)2 * c.dim * c.dim, c.dim, c.dim, c.dim, &vn, nullptr,
                                          stream))
                    return rc;
                L.w.ao_bound = 1.01f * l2 * vn;  // |softmax-weighted V| <= max |V_j| <= ||LN(x)|| ||Wv_j||
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

static int build_patch_operands(omnitok_engine *e, hipStream_t stream) {
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
        }
    }
    return OMNITOK_OK;
}

static int get_rope(omnitok_engine *e, int N, const float **cosp, const float **sinp, hipStream_t stream) {
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
static int get_bias_table(omnitok_engine *e, const std::string &prefix, int gh, int gw, const float **out,
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
// arithmetic of each kernel is independent of the problem / tile size (batch-size independence).
int g_gemm_mode = 2;
// Spatial attention arithmetic.  "attn_mode"
//   1 (default): fp16-split operands on the fp16 MFMA (attn_h2.hip) wherever the per-clip ranges of the row
//      statistics pass exist (the split-operand GEMM path) and the q/k scales are usable;
//   0: the fp32-input MFMA kernel of attn_spatial.hip.
int g_attn_mode = 1;
// "attn_vpack" 1 (default): the merged q|k|v launch writes V straight into the attention kernel's fp16 planes
int g_attn_vpack = 1;
// "gemm_pl" 1 (default): to_out / proj, FF-in and FF-out run as plane x plane GEMMs (gemm_pl.h) whose activation operands
// are written as fp16 hi|lo planes by their producers (attention kernels, the LayerNorm epilogue of to_out, the GEGLU
// epilogue of FF-in): no row-statistics pass and no in-loop LayerNorm / split in front of the FeedForward.  0: the
// r02 data flow (fp32 activations everywhere, gemm_h2.hip splits its A operand in the K loop).
int g_gemm_pl = 1;

static int gemm_mode_of(const omnitok_engine *e) { return e->opt_gemm_mode >= 0 ? e->opt_gemm_mode : g_gemm_mode; }
static int attn_mode_of(const omnitok_engine *e) { return e->opt_attn_mode >= 0 ? e->opt_attn_mode : g_attn_mode; }
static int attn_vpack_of(const omnitok_engine *e) { return e->opt_attn_vpack >= 0 ? e->opt_attn_vpack : g_attn_vpack; }
static int gemm_pl_of(const omnitok_engine *e) { return e->opt_gemm_pl >= 0 ? e->opt_gemm_pl : g_gemm_pl; }

static bool x3_ok(const omnitok_engine *e, int N, int K, int flags) {
    return gemm_mode_of(e) >= 1 && N % 32 == 0 && K % 32 == 0 && !(flags & OMNITOK_GEMM_LEAKY) &&
           (!(flags & OMNITOK_GEMM_GEGLU) || N % 64 == 0);
}

static float *next_bounds(omnitok_engine *e) {  // [n_clips][2] slots of one row-statistics launch
    if (e->bound_next >= N_BOUND_LAUNCHES) return nullptr;
    return e->bounds + (int64_t)2 * e->bound_clips * (e->bound_next++);
}

struct VPack {  // packed-V output of the merged q|k|v launch (gemm_h2.hip): planes, first V column, sequence shape, |v| bound
    void *planes; int col0, n_tokens, heads; float bound; const float *bound_dev;
};

static int eg_gemm(omnitok_engine *e, const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                   const float *residual, int64_t ldr, float *c, int64_t ldc, int64_t M, int N, int K, int flags,
                   int64_t rpg, int64_t gstride, int64_t goff, hipStream_t stream, ABound ab = ABound(),
                   const float *ln_stats = nullptr, const float *ln_g = nullptr, const float *ln_b = nullptr,
                   int ln_cols = 0, float ln_bound = 0.0f, float *c2 = nullptr, int64_t ldc2 = 0, int split_col = 0,
                   const VPack *vpk = nullptr, bool *vpacked = nullptr) {
    if (gemm_mode_of(e) == 2 && ab.stat > 0.0f && x3_ok(e, N, K, flags) && (!ln_stats || ln_bound > 0.0f) &&
        (!ab.dev || (ab.rpc > 0 && ab.rpc % 64 == 0 && rpg == 0))) {
        auto it = e->h2w.find(w);
        if (it != e->h2w.end() && ldw == K) {
            if (vpk && ab.dev && ab.rpc % 128 == 0) {  // V columns straight into the attention kernel's fp16 planes
                *vpacked = true;
                return omnitok_gemm_h2_vpack(a, lda, it->second.pl, it->second.sc, bias, residual, ldr, c, ldc, M, N, K,
                                             flags, rpg, gstride, goff, ab.stat, ab.dev, 2, ab.rpc, ln_stats, ln_g, ln_b,
                                             ln_cols, ln_bound, c2, ldc2, split_col, vpk->planes, vpk->col0, vpk->n_tokens,
                                             vpk->heads, vpk->bound, vpk->bound_dev, 2, stream);
            }
            return omnitok_gemm_h2(a, lda, it->second.pl, it->second.sc, bias, residual, ldr, c, ldc, M, N, K, flags,
                                   rpg, gstride, goff, ab.stat, ab.dev, 2, ab.rpc, ln_stats, ln_g, ln_b, ln_cols,
                                   ln_bound, c2, ldc2, split_col, stream);
        }
    }
    if (x3_ok(e, N, K, flags))
        return omnitok_gemm_x3(a, lda, w, ldw, bias, residual, ldr, c, ldc, M, N, K, flags, rpg, gstride, goff,
                               ln_stats, ln_g, ln_b, ln_cols, c2, ldc2, split_col, stream);
    if (ln_stats) {
        set_error("eg_gemm: fused LayerNorm needs the x3 / h2 kernel");
        return OMNITOK_ERR_STATE;
    }
    return omnitok_gemm(a, lda, w, ldw, bias, residual, ldr, c, ldc, M, N, K, flags, rpg, gstride, goff, stream);
}

// plane x plane GEMM of the engine: the weight is looked up by its fp32 pointer
static int eg_gemm_pl(omnitok_engine *e, omnitok_pl_gemm g, const float *w, hipStream_t stream) {
    auto it = e->plw.find(w);
    if (it == e->plw.end()) {
        set_error("eg_gemm_pl: weight was not packed for the plane GEMM");
        return OMNITOK_ERR_STATE;
    }
    g.w = it->second.pl;
    g.w_scale = it->second.sc;
    return omnitok_gemm_pl(&g, stream);
}

// One Transformer (reference attention.py:655-689). X holds the tokens on entry and on exit.
// Pooling blocks shrink the token grid (attention.py:683-684): *ghp / *gwp are updated.
// transpose_out: the final LayerNorm stores its rows in the OTHER stage's token order ('(b t)(h w)' <-> '(b h w) t'),
// i.e. the rearrange that follows every Transformer on the path is fused into the norm_out store.
static int run_transformer(omnitok_engine *e, const TransformerW &tw, int B, int T, int *ghp, int *gwp, bool spatial,
                           hipStream_t stream, bool transpose_out = false) {
    const omnitok_config &c = e->cfg;
    const int D = c.dim, heads = c.heads;
    int gh = *ghp, gw = *gwp;
    int64_t L = (int64_t)B * T * gh * gw;
    int S = gh * gw;
    double gemm_f = 2.0 * (double)L * D;
    // split-operand GEMMs apply the LayerNorm while staging their A operand (no LN pass over HBM)
    const bool fused = x3_ok(e, 3 * D, D, 0) && D <= 512;
    // Plane data flow (gemm_pl.h): attention output -> planes (AO) -> to_out / proj with the residual add and the
    // FeedForward's LayerNorm in its epilogue (x in place, LN(x) planes -> Y) -> FF-in with the GEGLU hidden as planes
    // (HD) -> FF-out (+ residual).  Needs full-row tiles for the LayerNorm epilogue (dim 512 = the reference's only width).
    const bool pl = gemm_pl_of(e) && gemm_mode_of(e) == 2 && fused && D == 512 && x3_ok(e, 2 * e->inner_pad, D, OMNITOK_GEMM_GEGLU);
    bool ln_planes_ready = false;  // Y holds the planes of the FeedForward's LayerNorm(x) for the current x
    // producers that cannot write planes themselves: fp32 rows -> planes with one power-of-two scale per row
    auto pack_rows = [&](const float *src, void *planes, float *scales) -> int {
        const int64_t Lp = (L + 255) / 256 * 256;
        OT_RUN("pack_rows", 2.0 * L * D * 4.0, omnitok_pl_pack_rows(src, D, L, D, Lp, planes, scales, 0.0f, stream));
        return OMNITOK_OK;
    };
    // to_out / proj as a plane GEMM: x += a . w^T (+ bias), and the planes of LN_ff(x) -> Y
    auto gemm_out_pl = [&](const Layer &ly, const void *a_planes, const float *a_scale, float a_const, const float *w,
                           const float *bias) -> int {
        omnitok_pl_gemm g{};
        g.a = a_planes;
        g.a_scale = a_scale;
        g.a_scale_const = a_const;
        g.bias = bias;
        g.residual = e->X.p;
        g.ldr = D;
        g.c = e->X.p;
        g.ldc = D;
        g.out_planes = e->Y.p;
        g.out_planes_k = D;
        g.out_bound = ly.ff.ln_bound;
        g.ln_gamma = ly.ff.lw;
        g.ln_beta = ly.ff.lb;
        g.ln_eps = 1e-5f;
        g.epilogue = 2;
        g.M = L;
        g.N = D;
        g.K = D;
        OT_RUN("gemm_out", gemm_f * D, eg_gemm_pl(e, g, w, stream));
        ln_planes_ready = true;
        return OMNITOK_OK;
    };
    for (const Layer &ly : tw.layers) {
        ln_planes_ready = false;
        if (ly.kind == 'a' || ly.kind == 'm' || ly.kind == 'l') {
            // Pooling (reference attention.py:83-113), no residual (:674); then FF (+residual) on the
            // quarter-size sequence
            if (!spatial || gh % 2 || gw % 2) {
                set_error("pooling block '%c' on a %dx%d grid", ly.kind, gh, gw);
                return OMNITOK_ERR_INVALID;
            }
            if (ly.kind == 'l')  // Linear(4D -> D) on four consecutive tokens: x.view(B, N/4, 4C)
                OT_RUN("pool", 2.0 * (double)L * D * D,
                       eg_gemm(e, e->X.p, 4 * D, ly.pool_w, 4 * D, ly.pool_b, nullptr, 0, e->X2.p, D, L / 4, D, 4 * D,
                               OMNITOK_GEMM_BIAS, 0, 0, 0, stream));
            else
                OT_RUN("pool", 1.25 * L * D * 4.0,
                       omnitok_token_resample(e->X.p, e->X2.p, ly.kind == 'a' ? 0 : 1, (int64_t)B * T, 1, gh, gw, D,
                                              stream));
            std::swap(e->X, e->X2);
            gh /= 2;
            gw /= 2;
            S = gh * gw;
            L = (int64_t)B * T * S;
            gemm_f = 2.0 * (double)L * D;
            goto feed_forward;
        }
        if (ly.kind == 'n' || ly.kind == 'r') {
            // Up (reference attention.py:116-150): nearest 2x2 up-sampling of the token grid, no residual (:674),
            // 'r' adds Linear(D, D).  A row-wise Linear commutes with the row duplication bit for bit, so it
            // runs on the L source rows (a quarter of the work) and the result is up-sampled.
            if (!spatial) {
                set_error("Up block '%c' in a temporal transformer", ly.kind);
                return OMNITOK_ERR_INVALID;
            }
            const float *src = e->X.p;
            if (ly.kind == 'r') {
                OT_RUN("pool", 2.0 * (double)L * D * D,
                       eg_gemm(e, e->X.p, D, ly.pool_w, D, ly.pool_b, nullptr, 0, e->Y.p, D, L, D, D, OMNITOK_GEMM_BIAS, 0,
                               0, 0, stream));
                src = e->Y.p;
            }
            OT_RUN("pool", 1.25 * 4.0 * L * D * 4.0,
                   omnitok_token_resample(src, e->X2.p, 2, (int64_t)B * T, 1, gh, gw, D, stream));
            std::swap(e->X, e->X2);
            gh *= 2;
            gw *= 2;
            S = gh * gw;
            L = (int64_t)B * T * S;
            gemm_f = 2.0 * (double)L * D;
            goto feed_forward;
        }
        if (ly.kind == 't') {
            OT_RUN("peg3d", 2.0 * L * D * 4.0,
                   omnitok_peg3d(e->X.p, ly.t.peg_w27, ly.t.peg_b, e->X2.p, B, T, gh, gw, D, c.causal_peg, stream));
            std::swap(e->X, e->X2);
            // device-side ranges of this layer's x (filled by the row-statistics pass): bs[0] >= max |x|,
            // bs[1] >= max ||x_row||; |V_j| <= ||x|| ||Wv_j|| bounds the attention output (a convex
            // combination of V rows) -- what the fp16-split GEMMs need (gemm_h2.hip)
            float *bs = fused ? next_bounds(e) : nullptr;
            if (fused && !bs) {
                set_error("run_transformer: out of range slots");
                return OMNITOK_ERR_STATE;
            }
            const int64_t rpc = L / B;  // both token orders keep a clip's rows contiguous (b is the outermost index)
            const ABound ab_x = bs ? ABound{1.01f, bs, rpc} : ABound();
            const ABound ab_ao = bs ? ABound{1.01f * ly.t.vnorm, bs + 1, rpc} : ABound();
            if (!spatial && T == 1) {
                // Images: a temporal sequence of one token.  softmax over a single key is exactly 1
                // (causal or not, with or without ALiBi), so the attention output is V bit for bit:
                // only the V half of to_kv is needed (rows [D, 2D) of the weight); LN, to_q, the K
                // half and the attention kernel drop out.  Identical results to the general path.
                if (bs && gemm_mode_of(e) == 2)  // only the ranges are needed here
                    OT_RUN("row_stats", L * D * 4.0, omnitok_row_stats(e->X.p, L, D, 1e-5f, e->ST.p, bs, rpc, stream));
                OT_RUN("gemm_qkv", gemm_f * D,
                       eg_gemm(e, e->X.p, D, ly.t.wkv + (int64_t)D * D, D, nullptr, nullptr, 0, e->AO.p, D, L, D, D, 0, 0,
                               0, 0, stream, ab_x));
                if (pl) {
                    if (int rc = pack_rows(e->AO.p, e->QKV.p, e->ST.p)) return rc;
                    if (int rc = gemm_out_pl(ly, e->QKV.p, e->ST.p, 1.0f, ly.t.wo, nullptr)) return rc;
                    goto feed_forward;
                }
                OT_RUN("gemm_out", gemm_f * D,
                       eg_gemm(e, e->AO.p, D, ly.t.wo, D, nullptr, e->X.p, D, e->X.p, D, L, D, D, OMNITOK_GEMM_RESIDUAL,
                               0, 0, 0, stream, ab_ao));
                goto feed_forward;
            }
            float *Q = e->QKV.p, *KV = e->QKV.p + L * D;
            int64_t ldq = D, ldkv = 2 * D;
            // fp16-split spatial attention (attn_h2.hip): Q, K (RoPE + l2norm + scales applied) and V as hi|lo planes
            // in MFMA fragment order (Y, HD are free here); |V| <= ||x_row|| ||Wv_j|| per clip from the row statistics
            const float qb = 1.01f * 8.0f * ly.t.q_amax, kb = 1.01f * ly.t.k_amax;
            const bool attn_h2 = spatial && attn_mode_of(e) == 1 && bs && S % 64 == 0 && qb > 0.0f && kb > 0.0f &&
                                 ly.t.vnorm > 0.0f && qb < 1e30f && kb < 1e30f;
            unsigned char *qp = reinterpret_cast<unsigned char *>(e->Y.p);
            unsigned char *kp = reinterpret_cast<unsigned char *>(e->HD.p);
            unsigned char *vp = kp + (size_t)L * D * 4;
            const VPack vpk{vp, 2 * D, S, heads, ab_ao.stat, ab_ao.dev};
            bool vpacked = false;  // the q|k|v launch wrote the V planes itself (no fp32 V round trip)
            bool ao_planes = false;  // the attention kernel wrote its output as planes (AO) with row scales (ST)
            // Q from LN(x), K/V from the raw x (reference attention.py:404-412)
            if (fused && D % 256 == 0) {
                // one launch on the merged weight: the LayerNorm is applied while the A tile is staged,
                // for the Q columns only; QKV rows are [q | k | v]
                OT_RUN("row_stats", L * D * 4.0, omnitok_row_stats(e->X.p, L, D, 1e-5f, e->ST.p, bs, rpc, stream));
                // ... and the two column ranges land in two dense tensors (Q [L, D], K|V [L, 2D]): the attention
                // kernels read rows of D / 2D floats (with a [L, 3D] row pitch spatial attention ran 15 % slower)
                OT_RUN("gemm_qkv", gemm_f * 3 * D,
                       eg_gemm(e, e->X.p, D, ly.t.wqkv, D, nullptr, nullptr, 0, Q, D, L, 3 * D, D, 0, 0, 0, 0, stream, ab_x,
                               e->ST.p, ly.t.ng, ly.t.nb, D, ly.t.ln_bound, KV, 2 * D, D, attn_h2 && attn_vpack_of(e) ? &vpk : nullptr,
                               &vpacked));
            } else if (fused) {
                OT_RUN("row_stats", L * D * 4.0, omnitok_row_stats(e->X.p, L, D, 1e-5f, e->ST.p, bs, rpc, stream));
                OT_RUN("gemm_qkv", gemm_f * D,
                       eg_gemm(e, e->X.p, D, ly.t.wq, D, nullptr, nullptr, 0, Q, D, L, D, D, 0, 0, 0, 0, stream, ab_x,
                               e->ST.p, ly.t.ng, ly.t.nb, D, ly.t.ln_bound));
                OT_RUN("gemm_qkv", gemm_f * 2 * D,
                       eg_gemm(e, e->X.p, D, ly.t.wkv, D, nullptr, nullptr, 0, KV, 2 * D, L, 2 * D, D, 0, 0, 0, 0, stream,
                               ab_x));
            } else {
                OT_RUN("layernorm", 2.0 * L * D * 4.0,
                       omnitok_layernorm(e->X.p, ly.t.ng, ly.t.nb, e->Y.p, L, D, 1e-5f, 0, 0, 0, stream));
                OT_RUN("gemm_qkv", gemm_f * D,
                       eg_gemm(e, e->Y.p, D, ly.t.wq, D, nullptr, nullptr, 0, Q, D, L, D, D, 0, 0, 0, 0, stream));
                OT_RUN("gemm_qkv", gemm_f * 2 * D,
                       eg_gemm(e, e->X.p, D, ly.t.wkv, D, nullptr, nullptr, 0, KV, 2 * D, L, 2 * D, D, 0, 0, 0, 0,
                               stream));
            }
            if (spatial) {
                const float *cosp = nullptr, *sinp = nullptr;
                if (c.spatial_rope)
                    if (int rc = get_rope(e, S, &cosp, &sinp, stream)) return rc;
                const float *bias = nullptr;
                if (!ly.t.bias_prefix.empty())
                    if (int rc = get_bias_table(e, ly.t.bias_prefix, gh, gw, &bias, stream)) return rc;
                if (attn_h2) {
                    OT_RUN("qk_prep", (vpacked ? 4.0 : 6.0) * L * D * 4.0,
                           omnitok_attn_pack(Q, ldq, KV, vpacked ? nullptr : KV + D, ldkv, L, S, heads, cosp, sinp,
                                             ly.t.q_scale, ly.t.k_scale, 8.0f, qb, kb, ab_ao.stat, ab_ao.dev, 2, rpc, qp, kp,
                                             vpacked ? nullptr : vp, stream));
                    OT_RUN("attn_spatial", 4.0 * (double)B * T * heads * (double)S * S * 64.0,
                           omnitok_attn_spatial_h2_planes(qp, kp, vp, e->AO.p, D, pl ? e->AO.p : nullptr, pl ? e->ST.p : nullptr,
                                                          B * T, S, heads, qb, kb, ab_ao.stat, ab_ao.dev, 2, T, bias, gh, gw,
                                                          stream));
                    ao_planes = pl;
                } else {
                    OT_RUN("qk_prep", 4.0 * L * D * 4.0,
                           omnitok_qk_prep(Q, ldq, KV, ldkv, L, S, heads, cosp, sinp, ly.t.q_scale, ly.t.k_scale, 8.0f,
                                           stream));
                    OT_RUN("attn_spatial", 4.0 * (double)B * T * heads * (double)S * S * 64.0,
                           omnitok_attn_spatial(Q, ldq, KV, KV + D, ldkv, e->AO.p, D, B * T, S, heads, bias, gh, gw,
                                                stream));
                }
            } else {
                const float *alibi = (c.legacy_attention && c.causal_temporal) ? e->alibi : nullptr;
                const bool tp = pl && bs && T <= 17 && S % 16 == 0;
                OT_RUN("attn_temporal", 4.0 * L * D * 4.0,
                       omnitok_attn_temporal_planes(Q, ldq, KV, KV + D, ldkv, e->AO.p, D, tp ? e->AO.p : nullptr,
                                                    tp ? e->ST.p : nullptr, ab_ao.stat, ab_ao.dev, 2, S, (int64_t)B * S, T,
                                                    heads, ly.t.q_scale, ly.t.k_scale, 8.0f, c.causal_temporal, alibi,
                                                    stream));
                ao_planes = tp;
            }
            if (pl) {
                const void *ap = e->AO.p;
                if (!ao_planes) {  // fp32 attention output (fp32-MFMA attention kernel, long temporal sequences): pack it
                    if (int rc = pack_rows(e->AO.p, e->QKV.p, e->ST.p)) return rc;
                    ap = e->QKV.p;
                }
                if (int rc = gemm_out_pl(ly, ap, e->ST.p, 1.0f, ly.t.wo, nullptr)) return rc;
                goto feed_forward;
            }
            OT_RUN("gemm_out", gemm_f * D,
                   eg_gemm(e, e->AO.p, D, ly.t.wo, D, nullptr, e->X.p, D, e->X.p, D, L, D, D, OMNITOK_GEMM_RESIDUAL, 0, 0,
                           0, stream, ab_ao));
        } else {
            if (fused) {
                OT_RUN("row_stats", L * D * 4.0, omnitok_row_stats(e->X.p, L, D, 1e-5f, e->ST.p, nullptr, 0, stream));
                OT_RUN("gemm_qkv", gemm_f * 3 * D,
                       eg_gemm(e, e->X.p, D, ly.w.wqkv, D, nullptr, nullptr, 0, e->QKV.p, 3 * D, L, 3 * D, D, 0, 0, 0, 0,
                               stream, ABound{1.0f, nullptr, 0}, e->ST.p, ly.w.ng, ly.w.nb, 3 * D, ly.w.ln_bound));
            } else {
                OT_RUN("layernorm", 2.0 * L * D * 4.0,
                       omnitok_layernorm(e->X.p, ly.w.ng, ly.w.nb, e->Y.p, L, D, 1e-5f, 0, 0, 0, stream));
                OT_RUN("gemm_qkv", gemm_f * 3 * D,
                       eg_gemm(e, e->Y.p, D, ly.w.wqkv, D, nullptr, nullptr, 0, e->QKV.p, 3 * D, L, 3 * D, D, 0, 0, 0, 0,
                               stream));
            }
            if (pl && ly.w.ao_bound > 0.0f) {
                OT_RUN("attn_window", 4.0 * (double)L * 64.0 * D,
                       omnitok_attn_window_planes(e->QKV.p, 3 * D, ly.w.bias_dense, nullptr, 0, e->AO.p, ly.w.ao_bound, B * T, gh,
                                                  gw, heads, stream));
                if (int rc = gemm_out_pl(ly, e->AO.p, nullptr, omnitok_pl_unscale(ly.w.ao_bound), ly.w.wproj, ly.w.bproj))
                    return rc;
                goto feed_forward;
            }
            OT_RUN("attn_window", 4.0 * (double)L * 64.0 * D,
                   omnitok_attn_window(e->QKV.p, 3 * D, ly.w.bias_dense, e->AO.p, D, B * T, gh, gw, heads, stream));
            OT_RUN("gemm_out", gemm_f * D,
                   eg_gemm(e, e->AO.p, D, ly.w.wproj, D, ly.w.bproj, e->X.p, D, e->X.p, D, L, D, D,
                           OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL, 0, 0, 0, stream,
                           fused ? ABound{ly.w.ao_bound, nullptr, 0} : ABound()));
        }
    feed_forward:
        // FeedForward (reference attention.py:153-168)
        if (pl) {
            if (!ln_planes_ready) {  // after pooling / Up blocks: LayerNorm pass, then planes
                const int64_t Lp = (L + 255) / 256 * 256;
                OT_RUN("layernorm", 2.0 * L * D * 4.0,
                       omnitok_layernorm(e->X.p, ly.ff.lw, ly.ff.lb, e->AO.p, L, D, 1e-5f, 0, 0, 0, stream));
                OT_RUN("pack_rows", 2.0 * L * D * 4.0,
                       omnitok_pl_pack_rows(e->AO.p, D, L, D, Lp, e->Y.p, nullptr, ly.ff.ln_bound, stream));
            }
            omnitok_pl_gemm g{};
            g.a = e->Y.p;
            g.a_scale_const = omnitok_pl_unscale(ly.ff.ln_bound);
            g.out_planes = e->HD.p;
            g.out_planes_k = e->inner_pad;
            g.out_bound = ly.ff.h_bound;
            g.epilogue = 1;
            g.M = L;
            g.N = 2 * e->inner_pad;
            g.K = D;
            OT_RUN("gemm_ff_in", gemm_f * 2 * c.ff_inner, eg_gemm_pl(e, g, ly.ff.w1p, stream));
            omnitok_pl_gemm h{};
            h.a = e->HD.p;
            h.a_scale_const = omnitok_pl_unscale(ly.ff.h_bound);
            h.residual = e->X.p;
            h.ldr = D;
            h.c = e->X.p;
            h.ldc = D;
            h.epilogue = 0;
            h.M = L;
            h.N = D;
            h.K = e->inner_pad;
            OT_RUN("gemm_ff_out", gemm_f * c.ff_inner, eg_gemm_pl(e, h, ly.ff.w2p, stream));
        } else if (fused && x3_ok(e, 2 * e->inner_pad, D, OMNITOK_GEMM_GEGLU)) {
            OT_RUN("row_stats", L * D * 4.0, omnitok_row_stats(e->X.p, L, D, 1e-5f, e->ST.p, nullptr, 0, stream));
            OT_RUN("gemm_ff_in", gemm_f * 2 * c.ff_inner,
                   eg_gemm(e, e->X.p, D, ly.ff.w1p, D, nullptr, nullptr, 0, e->HD.p, e->inner_pad, L, 2 * e->inner_pad, D,
                           OMNITOK_GEMM_GEGLU, 0, 0, 0, stream, ABound{1.0f, nullptr, 0}, e->ST.p, ly.ff.lw, ly.ff.lb,
                           2 * e->inner_pad, ly.ff.ln_bound));
            OT_RUN("gemm_ff_out", gemm_f * c.ff_inner,
                   eg_gemm(e, e->HD.p, e->inner_pad, ly.ff.w2p, e->inner_pad, nullptr, e->X.p, D, e->X.p, D, L, D,
                           e->inner_pad, OMNITOK_GEMM_RESIDUAL, 0, 0, 0, stream, ABound{ly.ff.h_bound, nullptr, 0}));
        } else {
            OT_RUN("layernorm", 2.0 * L * D * 4.0,
                   omnitok_layernorm(e->X.p, ly.ff.lw, ly.ff.lb, e->Y.p, L, D, 1e-5f, 0, 0, 0, stream));
            OT_RUN("gemm_ff_in", gemm_f * 2 * c.ff_inner,
                   eg_gemm(e, e->Y.p, D, ly.ff.w1p, D, nullptr, nullptr, 0, e->HD.p, e->inner_pad, L, 2 * e->inner_pad,
                           D, OMNITOK_GEMM_GEGLU, 0, 0, 0, stream));
            OT_RUN("gemm_ff_out", gemm_f * c.ff_inner,
                   eg_gemm(e, e->HD.p, e->inner_pad, ly.ff.w2p, e->inner_pad, nullptr, e->X.p, D, e->X.p, D, L, D,
                           e->inner_pad, OMNITOK_GEMM_RESIDUAL, 0, 0, 0, stream));
        }
    }
    if (transpose_out && T > 1)  // rows (b, t, s) -> (b, s, t) after a spatial stage, (b, s, t) -> (b, t, s) after a temporal one
        OT_RUN("layernorm", 2.0 * L * D * 4.0,
               omnitok_layernorm_transposed(e->X.p, tw.og, tw.ob, e->X2.p, B, spatial ? T : S, spatial ? S : T, D, 1e-5f,
                                            stream));
    else
        OT_RUN("layernorm", 2.0 * L * D * 4.0,
               omnitok_layernorm(e->X.p, tw.og, tw.ob, e->X2.p, L, D, 1e-5f, 0, 0, 0, stream));
    std::swap(e->X, e->X2);
    *ghp = gh;
    *gwp = gw;
    return OMNITOK_OK;
}

// range slots of the fp16-split GEMMs: per clip, zeroed once per encode / decode (atomic max targets)
static int reset_bounds(omnitok_engine *e, int B, hipStream_t stream) {
    if (B > e->bound_cap) {
        // grow-only, and the old block stays allocated until the engine dies: a HIP graph captured at a smaller batch has
        // its address baked in (freeing it would hand the replay recycled memory).  512 B per clip: sized generously.
        if (e->bounds) e->retired.push_back(e->bounds);
        e->bounds = nullptr;
        const int cap = B > 1024 ? B : 1024;
        OT_HIP(hipMalloc(reinterpret_cast<void **>(&e->bounds), (size_t)2 * cap * N_BOUND_LAUNCHES * sizeof(float)));
        e->bound_cap = cap;
    }
    e->bound_clips = B;
    e->bound_next = 0;
    if (int rc = device_fill_u32(e->bounds, 0u, (int64_t)2 * B * N_BOUND_LAUNCHES, stream)) return rc;
    return OMNITOK_OK;
}

// floats per token of the eight workspace buffers, in the order X, X2, Y, QKV, AO, HD, Z, ST
static void workspace_widths(const omnitok_engine *e, int64_t (&wd)[8]) {
    const omnitok_config &c = e->cfg;
    const int D = c.dim;
    const Geo g = geometry(c);
    const int kenc = ((c.image_channels * g.p_enc * g.p_enc * g.pt_enc + 31) / 32) * 32;
    const int kdec = c.image_channels * g.p_dec * g.p_dec * g.pt_dec;
    const int kmax = kenc > kdec ? kenc : kdec;
    int64_t hdw = e->inner_pad;
    if (kmax > hdw) hdw = kmax;
    if (2 * D > hdw) hdw = 2 * D;  // packed K | V of the fp16-split attention
    const int64_t w[8] = {D, D, D, 3 * (int64_t)D, D, hdw, 8, 2};
    for (int i = 0; i < 8; ++i) wd[i] = w[i];
}

static int64_t workspace_bytes_for(const omnitok_engine *e, int64_t L) {
    int64_t wd[8], total = 0;
    workspace_widths(e, wd);
    L = (L + 255) / 256 * 256;  // plane operands are read in whole 256-row tiles (gemm_pl.h)
    for (int i = 0; i < 8; ++i) total += ((L * wd[i] * 4 + 255) / 256) * 256;
    return total;
}

static int ensure_workspace(omnitok_engine *e, int64_t L) {
    int64_t wd[8];
    workspace_widths(e, wd);
    const int64_t L_real = L;
    L = (L + 255) / 256 * 256;
    Buf *bufs[8] = {&e->X, &e->X2, &e->Y, &e->QKV, &e->AO, &e->HD, &e->Z, &e->ST};
    if (e->ext_ws) {  // slices of the caller's block, 256-byte aligned
        const int64_t need = workspace_bytes_for(e, L);
        if (need > e->ext_ws_bytes) {
            set_error("workspace of %lld bytes is too small: %lld tokens need %lld (omnitok_engine_workspace_need_*)",
                      (long long)e->ext_ws_bytes, (long long)L_real, (long long)need);
            return OMNITOK_ERR_STATE;
        }
        char *p = reinterpret_cast<char *>(e->ext_ws);
        for (int i = 0; i < 8; ++i) {
            bufs[i]->p = reinterpret_cast<float *>(p);
            bufs[i]->cap = L * wd[i];
            p += ((L * wd[i] * 4 + 255) / 256) * 256;
        }
        return OMNITOK_OK;
    }
    for (int i = 0; i < 8; ++i)
        if (int rc = ensure(*bufs[i], L * wd[i])) return rc;
    return OMNITOK_OK;
}

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
static int encode_shape(omnitok_engine *e, int F, int H, int W_, int *T, int *gh, int *gw) {
    const omnitok_config &c = e->cfg;
    const Geo g = geometry(c);
    // reference omnitokenizer.py:931-932
    OT_CHECK_ARG(F >= 1 && (F - 1) % g.pt_enc == 0,
                 "number of frames (%d) minus one (%d) must be divisible by temporal patch size (%d)", F, F - 1,
                 g.pt_enc);
    OT_CHECK_ARG(H > 0 && W_ > 0 && H % g.p_enc == 0 && W_ % g.p_enc == 0,
                 "image size %dx%d not divisible by patch size %d", H, W_, g.p_enc);
    int t = 1 + (F - 1) / g.pt_enc, h = H / g.p_enc, w = W_ / g.p_enc;
    OT_CHECK_ARG(walk_enc_grid(c.enc_block, &h, &w, nullptr) && (!g.defer_s || (h % 2 == 0 && w % 2 == 0)),
                 "token grid %dx%d not divisible by the pooling factor of enc_block '%s'%s", H / g.p_enc, W_ / g.p_enc,
                 c.enc_block, g.defer_s ? " + deferred spatial pool" : "");
    if (g.defer_s) {
        h /= 2;
        w /= 2;
    }
    if (g.defer_t) t = 1 + (t - 1) / 2;  // AvgPool3d((2,1,1)) floors, omnitokenizer.py:909-914
    *T = t;
    *gh = h;
    *gw = w;
    return OMNITOK_OK;
}

static int decode_shape(omnitok_engine *e, int T, int gh, int gw, int *F, int *H, int *W_) {
    const Geo g = geometry(e->cfg);
    OT_CHECK_ARG(T >= 1 && gh >= 1 && gw >= 1, "decode: bad latent shape %dx%dx%d", T, gh, gw);
    const int t = g.defer_t ? 1 + (T - 1) * 2 : T, up = g.defer_s ? 2 : 1;
    *F = 1 + (t - 1) * g.pt_dec;
    *H = gh * up * g.p_dec;
    *W_ = gw * up * g.p_dec;
    return OMNITOK_OK;
}

static int check_attention_grid(const char *who, const std::string &block, int gh, int gw) {
    // grids seen by the blocks of a spatial transformer ('t': N % 64, square; 'w': 8x8 windows)
    for (char ch : block) {
        if (ch == 't') {
            OT_CHECK_ARG((gh * gw) % 64 == 0, "%s: %d tokens per frame; spatial attention needs a multiple of 64", who,
                         gh * gw);
            OT_CHECK_ARG(gh == gw, "%s: the reference assumes a square token grid (int(sqrt(N)), attention.py:261)", who);
        } else if (ch == 'w') {
            OT_CHECK_ARG(gh % 8 == 0 && gw % 8 == 0 && gh == gw, "%s: token grid %dx%d not divisible by the 8x8 window",
                         who, gh, gw);
        } else if (ch == 'n' || ch == 'r') {
            OT_CHECK_ARG(gh == gw, "%s: Up block on a %dx%d grid (the reference assumes a square grid, attention.py:141)",
                         who, gh, gw);
            gh *= 2;
            gw *= 2;
        } else {
            OT_CHECK_ARG(gh % 2 == 0 && gw % 2 == 0 && gh == gw, "%s: pooling block on a %dx%d grid", who, gh, gw);
            gh /= 2;
            gw /= 2;
        }
    }
    return OMNITOK_OK;
}

// encoder up to the pre_vq input: leaves tokens [B, T', h, w, D] in e->X  (reference
// omnitokenizer.py:881-947 OmniTokenizer_Encoder.forward / encode)
static int encode_tokens(omnitok_engine *e, const float *x, int B, int F, int H, int W_, int *T_out, int *S_out,
                         hipStream_t stream) {
    if (!e->finalized) {
        set_error("encode: engine not finalised (load the weights first)");
        return OMNITOK_ERR_STATE;
    }
    const omnitok_config &c = e->cfg;
    const Geo g = geometry(c);
    const int D = c.dim, C = c.image_channels, p = g.p_enc, pt = g.pt_enc;
    int To, gho, gwo;
    if (int rc = encode_shape(e, F, H, W_, &To, &gho, &gwo)) return rc;
    int gh = H / p, gw = W_ / p;
    const int S = gh * gw, T = 1 + (F - 1) / pt;
    if (int rc = check_attention_grid("encode", c.enc_block, gh, gw)) return rc;
    const int64_t L = (int64_t)B * T * S;
    {
        int ph = gh, pw = gw;
        int64_t peak = S;  // Up blocks grow the sequence: size the workspace for the largest grid on the way
        (void)walk_enc_grid(c.enc_block, &ph, &pw, &peak);
        if (int rc = ensure_workspace(e, (int64_t)B * T * peak)) return rc;
    }
    if (int rc = reset_bounds(e, B, stream)) return rc;
    const char *names[2] = {"encoder.to_patch_emb_first_frame", "encoder.to_patch_emb"};

    // ---- patch embedding (reference omnitokenizer.py:806-838, 934-945) ----------------------
    for (int i = 0; i < 2; ++i) {
        const int tt = i ? T - 1 : 1, f0 = i, pti = i ? pt : 1;
        if (tt == 0) break;
        const int64_t M = (int64_t)B * tt * S;
        const std::string n = names[i];
        const int K = e->pe_k[i], ld = e->pe_ld[i];
        // rows (b, t, s) of this frame group -> rows of the [B, T, S] token tensor
        const int64_t rpg = (int64_t)tt * S, gstride = (int64_t)T * S, goff = i ? S : 0;
        if (c.patch_embed_cnn) {
            // Conv3d with kernel == stride is im2col x W; eval-mode BatchNorm is folded into W, b
            OT_RUN("patchify_ln", ((double)M * K) * 8.0,
                   omnitok_patchify_ln(x, B, C, F, H, W_, f0, tt, pti, p, nullptr, nullptr, 0.0f, e->HD.p, ld, stream));
            OT_RUN("gemm_patch", 2.0 * M * (double)K * D,
                   eg_gemm(e, e->HD.p, ld, e->pe_w[i], ld, e->pe_b[i], nullptr, 0, e->AO.p, D, M, D, ld,
                           OMNITOK_GEMM_BIAS, 0, 0, 0, stream));
            // scatter the frame group into the token tensor (a strided row copy)
            for (int b = 0; b < B; ++b)
                OT_HIP(hipMemcpyAsync(e->X.p + ((int64_t)b * gstride + goff) * D, e->AO.p + (int64_t)b * rpg * D,
                                      (size_t)rpg * D * 4, hipMemcpyDeviceToDevice, stream));
            continue;
        }
        OT_RUN("patchify_ln", ((double)M * K) * 8.0,
               omnitok_patchify_ln(x, B, C, F, H, W_, f0, tt, pti, p, W(e, n + ".1.weight"), W(e, n + ".1.bias"), 1e-5f,
                                   e->HD.p, ld, stream));
        OT_RUN("gemm_patch", 2.0 * M * (double)K * D,
               eg_gemm(e, e->HD.p, ld, e->pe_w[i], ld, e->pe_b[i], nullptr, 0, e->AO.p, D, M, D, ld, OMNITOK_GEMM_BIAS,
                       0, 0, 0, stream, ABound{e->pe_bound[i], nullptr, 0}));
        OT_RUN("layernorm", 2.0 * M * D * 4.0,
               omnitok_layernorm(e->AO.p, W(e, n + ".3.weight"), W(e, n + ".3.bias"), e->X.p, M, D, 1e-5f, rpg, gstride,
                                 goff, stream));
    }
    // ---- spatial then temporal transformer (reference omnitokenizer.py:891-903) -------------
    // the rearranges '(b t)(h w) d -> (b h w) t d' and back (omnitokenizer.py:900, 906) ride on the norm_out stores
    if (int rc = run_transformer(e, e->enc_s, B, T, &gh, &gw, true, stream, true)) return rc;
    const int S2 = gh * gw;  // pooling blocks shrink the grid, omnitokenizer.py:898-899
    const int64_t L2 = (int64_t)B * T * S2;
    if (int rc = run_transformer(e, e->enc_t, B, T, &gh, &gw, false, stream, true)) return rc;
    // ---- deferred pools (reference omnitokenizer.py:907-914) ---------------------------------
    if (g.defer_s) {
        OT_RUN("pool", 1.25 * L2 * D * 4.0,
               omnitok_token_resample(e->X.p, e->X2.p, 0, (int64_t)B * T, 1, gh, gw, D, stream));
        std::swap(e->X, e->X2);
        gh /= 2;
        gw /= 2;
    }
    if (g.defer_t && T > 1) {
        OT_RUN("pool", 1.5 * (double)B * T * gh * gw * D * 4.0,
               omnitok_token_resample(e->X.p, e->X2.p, 3, B, T, gh * gw, 1, D, stream));
        std::swap(e->X, e->X2);
    }
    if (gh != gho || gw != gwo) {
        set_error("encode: internal shape mismatch");
        return OMNITOK_ERR_STATE;
    }
    *T_out = To;
    *S_out = gh * gw;
    return OMNITOK_OK;
}

extern "C" int omnitok_encode(omnitok_engine *e, const float *x, int B, int F, int H, int W_, int64_t *ids_out,
                              float *emb_out, float *z_out, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(e, "encode: null engine");
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG(x && ids_out, "encode: null pointer");
    if (e->cfg.use_vae) {
        set_error("encode: engine was built with use_vae (no quantiser on the path, reference "
                  "omnitokenizer.py:260-266); call omnitok_encode_vae");
        return OMNITOK_ERR_STATE;
    }
    int T = 0, S = 0;
    if (int rc = encode_tokens(e, x, B, F, H, W_, &T, &S, stream)) return rc;
    const omnitok_config &c = e->cfg;
    const int D = c.dim;
    const int64_t L = (int64_t)B * T * S;
    // ---- pre_vq + l2norm + nearest code (reference omnitokenizer.py:248-255) ----------------
    OT_RUN("pre_vq", (double)L * D * 4.0,
           omnitok_pre_vq(e->X.p, W(e, k_pre_w(c)), W(e, k_pre_b(c)), e->Z.p, L, D, 8, c.l2_code, stream));
    if (c.external_codebook) {
        // cosine similarity (l2_code): first argmax of the dot (vector_quantize_pytorch.py:646-650);
        // otherwise EuclideanCodebook: first argmax of -cdist (:463)
        if (c.l2_code)
            OT_RUN("vq_argmin", 2.0 * (double)L * c.n_codes * 8.0,
                   omnitok_vq_argmax_cos(e->Z.p, e->cb_packed, L, c.n_codes, ids_out, stream));
        else
            OT_RUN("vq_argmin", 2.0 * (double)L * c.n_codes * 8.0,
                   omnitok_vq_argmin_cdist(e->Z.p, e->cb_packed, e->cb_ee, L, c.n_codes, ids_out, stream));
        if (emb_out)  // eval: quantize = embed[ids] -> project_out, no straight-through term; [B,T',h,w,dim]
            OT_RUN("dequant_post_vq", (double)L * D * 4.0,
                   omnitok_gather_rows(ids_out, e->dq_table, c.n_codes, emb_out, L, D, nullptr, stream));
    } else {
        OT_RUN("vq_argmin", 2.0 * (double)L * c.n_codes * 8.0,
               omnitok_vq_argmin(e->Z.p, e->cb_packed, e->cb_ee, L, c.n_codes, ids_out, stream));
        if (emb_out)
            OT_RUN("vq_embed_st", (double)L * 8 * 8.0,
                   omnitok_vq_embed_st(ids_out, e->Z.p, W(e, k_embed(c)), 8, B, (int64_t)T * S, emb_out, stream));
    }
    if (z_out) OT_HIP(hipMemcpyAsync(z_out, e->Z.p, (size_t)L * 8 * 4, hipMemcpyDeviceToDevice, stream));
    return OMNITOK_OK;
}

enum class LatentKind { Ids, ChannelLast, ChannelFirst };

// shared decode body: latent (ids or continuous z) -> post_vq -> [deferred up-sampling] -> temporal ->
// spatial -> to_pixels  (reference omnitokenizer.py:1101-1118 forward, :1059-1098 decode)
static int decode_latent(omnitok_engine *e, LatentKind kind, const void *latent, int B, int T, int gh, int gw,
                         float *pixels_out, hipStream_t stream) {
    if (!e->finalized) {
        set_error("decode: engine not finalised (load the weights first)");
        return OMNITOK_ERR_STATE;
    }
    const omnitok_config &c = e->cfg;
    const Geo g = geometry(c);
    const int p = g.p_dec, pt = g.pt_dec, D = c.dim, C = c.image_channels;
    int F, H, W_;
    if (int rc = decode_shape(e, T, gh, gw, &F, &H, &W_)) return rc;
    const int T2 = g.defer_t ? 1 + (T - 1) * 2 : T, gh2 = g.defer_s ? gh * 2 : gh, gw2 = g.defer_s ? gw * 2 : gw;
    if (int rc = check_attention_grid("decode", std::string("t") + c.dec_block, gh2, gw2)) return rc;
    const int64_t L0 = (int64_t)B * T * gh * gw;
    const int S = gh2 * gw2;
    const int64_t L = (int64_t)B * T2 * S;
    if (int rc = ensure_workspace(e, L)) return rc;
    if (int rc = reset_bounds(e, B, stream)) return rc;
    const int K0 = C * p * p, K1 = K0 * pt;

    if (kind == LatentKind::Ids) {
        // the out-of-range flag reports on THIS decode only (omnitok_engine_check_ids)
        if (int rc = device_fill_u32(e->err_flag, 0u, 1, stream)) return rc;
    }
    // ids straight into the temporal stage's '(b h w) t' order when nothing sits between (no deferred up-sampling)
    const bool fuse_first_transpose = kind == LatentKind::Ids && !g.defer_t && !g.defer_s && T > 1;
    if (kind == LatentKind::Ids)
        OT_RUN("dequant_post_vq", (double)L0 * D * 4.0,
               omnitok_gather_rows_transposed(static_cast<const int64_t *>(latent), e->dq_table, c.n_codes, e->X.p, L0,
                                              fuse_first_transpose ? T : 0, fuse_first_transpose ? gh * gw : 0, D,
                                              e->err_flag, stream));
    else
        OT_RUN("post_vq", (double)L0 * D * 4.0,
               omnitok_post_vq(static_cast<const float *>(latent), kind == LatentKind::ChannelFirst, B,
                               (int64_t)T * gh * gw, 8, W(e, "post_vq_conv.1.weight"), W(e, "post_vq_conv.1.bias"),
                               e->X.p, D, stream));
    // deferred up-sampling: time first, then space (reference omnitokenizer.py:1103-1109)
    if (g.defer_t && T > 1) {
        OT_RUN("pool", 1.5 * (double)B * T2 * gh * gw * D * 4.0,
               omnitok_token_resample(e->X.p, e->X2.p, 4, B, T, gh * gw, 1, D, stream));
        std::swap(e->X, e->X2);
    }
    if (g.defer_s) {
        OT_RUN("pool", 1.25 * L * D * 4.0,
               omnitok_token_resample(e->X.p, e->X2.p, 2, (int64_t)B * T2, 1, gh, gw, D, stream));
        std::swap(e->X, e->X2);
    }
    int ghc = gh2, gwc = gw2;
    // temporal first on decode (reference omnitokenizer.py:1072-1084)
    if (T2 > 1 && !fuse_first_transpose) {
        OT_RUN("transpose", 2.0 * L * D * 4.0, omnitok_transpose_tokens(e->X.p, e->X2.p, B, T2, S, D, stream));
        std::swap(e->X, e->X2);
    }
    if (int rc = run_transformer(e, e->dec_t, B, T2, &ghc, &gwc, false, stream, true)) return rc;
    if (int rc = run_transformer(e, e->dec_s, B, T2, &ghc, &gwc, true, stream)) return rc;
    // ---- to_pixels (reference omnitokenizer.py:1006-1033, 1089-1096) -------------------------
    OT_RUN("gemm_pixels", 2.0 * B * S * (double)K0 * D,
           eg_gemm(e, e->X.p, D, e->px_w[0], D, e->px_b[0], nullptr, 0, e->HD.p, K0, (int64_t)B * S, K0, D,
                   OMNITOK_GEMM_BIAS, S, (int64_t)T2 * S, 0, stream, ABound{e->dec_s.out_bound, nullptr, 0}));
    OT_RUN("unpatchify", (double)B * S * K0 * 8.0,
           omnitok_unpatchify(e->HD.p, B, C, F, H, W_, 0, 1, 1, p, pixels_out, stream));
    if (T2 > 1) {
        const int64_t M1 = (int64_t)B * (T2 - 1) * S;
        OT_RUN("gemm_pixels", 2.0 * M1 * (double)K1 * D,
               eg_gemm(e, e->X.p, D, e->px_w[1], D, e->px_b[1], nullptr, 0, e->HD.p, K1, M1, K1, D, OMNITOK_GEMM_BIAS,
                       (int64_t)(T2 - 1) * S, (int64_t)T2 * S, S, stream, ABound{e->dec_s.out_bound, nullptr, 0}));
        OT_RUN("unpatchify", (double)M1 * K1 * 8.0,
               omnitok_unpatchify(e->HD.p, B, C, F, H, W_, 1, T2 - 1, pt, p, pixels_out, stream));
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_engine_encode_shape(omnitok_engine *e, int F, int H, int W_, int *T, int *h, int *w) {
    OT_CHECK_ARG(e && T && h && w, "encode_shape: null pointer");
    return encode_shape(e, F, H, W_, T, h, w);
}

extern "C" int omnitok_engine_decode_shape(omnitok_engine *e, int T, int h, int w, int *F, int *H, int *W_) {
    OT_CHECK_ARG(e && F && H && W_, "decode_shape: null pointer");
    return decode_shape(e, T, h, w, F, H, W_);
}

extern "C" int omnitok_decode(omnitok_engine *e, const int64_t *ids, int B, int T, int gh, int gw, float *pixels_out,
                              omnitok_stream_t stream_) {
    OT_CHECK_ARG(e, "decode: null engine");
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG(ids && pixels_out, "decode: null pointer");
    if (e->cfg.use_vae) {
        set_error("decode: engine was built with use_vae (latents are continuous, reference "
                  "omnitokenizer.py:293-317); call omnitok_decode_vae");
        return OMNITOK_ERR_STATE;
    }
    return decode_latent(e, LatentKind::Ids, ids, B, T, gh, gw, pixels_out, static_cast<hipStream_t>(stream_));
}

extern "C" int omnitok_encode_vae(omnitok_engine *e, const float *x, int B, int F, int H, int W_, const float *noise,
                                  float *z_out, float *moments_out, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(e, "encode_vae: null engine");
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG(x && z_out, "encode_vae: null pointer");
    if (!e->cfg.use_vae) {
        set_error("encode_vae: engine was built without use_vae; call omnitok_encode");
        return OMNITOK_ERR_STATE;
    }
    int T = 0, S = 0;
    if (int rc = encode_tokens(e, x, B, F, H, W_, &T, &S, stream)) return rc;
    const int D = e->cfg.dim;
    const int64_t L = (int64_t)B * T * S;
    OT_RUN("vae_sample", (double)L * D * 4.0,
           omnitok_vae_sample(e->X.p, W(e, "pre_vq_conv.1.weight"), W(e, "pre_vq_conv.1.bias"), noise, z_out,
                              moments_out, B, (int64_t)T * S, D, 8, stream));
    return OMNITOK_OK;
}

extern "C" int omnitok_decode_vae(omnitok_engine *e, const float *z, int channel_first, int B, int T, int gh, int gw,
                                  float *pixels_out, omnitok_stream_t stream_) {
    OT_CHECK_ARG(e, "decode_vae: null engine");
    if (B == 0) return OMNITOK_OK;
    OT_CHECK_ARG(z && pixels_out, "decode_vae: null pointer");
    if (!e->cfg.use_vae) {
        set_error("decode_vae: engine was built without use_vae; call omnitok_decode");
        return OMNITOK_ERR_STATE;
    }
    return decode_latent(e, channel_first ? LatentKind::ChannelFirst : LatentKind::ChannelLast, z, B, T, gh, gw,
                         pixels_out, static_cast<hipStream_t>(stream_));
}

// tokens the workspace must hold for an encode of [B, C, F, H, W] / a decode of [B, T, h, w] latents
static int64_t encode_peak_tokens(omnitok_engine *e, int B, int F, int H, int W_) {
    const Geo g = geometry(e->cfg);
    if (F < 1 || H % g.p_enc || W_ % g.p_enc || (F - 1) % g.pt_enc) return -1;
    int gh = H / g.p_enc, gw = W_ / g.p_enc;
    int64_t peak = (int64_t)gh * gw;
    (void)walk_enc_grid(e->cfg.enc_block, &gh, &gw, &peak);
    return (int64_t)B * (1 + (F - 1) / g.pt_enc) * peak;
}

extern "C" int64_t omnitok_engine_workspace_need_encode(omnitok_engine *e, int B, int F, int H, int W_) {
    if (!e) return -1;
    const int64_t L = encode_peak_tokens(e, B, F, H, W_);
    return L < 0 ? -1 : workspace_bytes_for(e, L);
}

extern "C" int64_t omnitok_engine_workspace_need_decode(omnitok_engine *e, int B, int T, int gh, int gw) {
    if (!e || T < 1 || gh < 1 || gw < 1) return -1;
    const Geo g = geometry(e->cfg);
    const int T2 = g.defer_t ? 1 + (T - 1) * 2 : T, up = g.defer_s ? 2 : 1;
    return workspace_bytes_for(e, (int64_t)B * T2 * gh * up * gw * up);
}

extern "C" int omnitok_engine_set_workspace(omnitok_engine *e, void *dev_ptr, int64_t bytes) {
    OT_CHECK_ARG(e, "set_workspace: null engine");
    OT_CHECK_ARG((dev_ptr == nullptr) == (bytes == 0) && bytes >= 0, "set_workspace: pointer / size mismatch");
    OT_CHECK_ARG((reinterpret_cast<uintptr_t>(dev_ptr) & 255) == 0, "set_workspace: the block must be 256-byte aligned");
    Buf *bufs[8] = {&e->X, &e->X2, &e->Y, &e->QKV, &e->AO, &e->HD, &e->Z, &e->ST};
    if (!e->ext_ws)  // leaving the internal allocation: release it
        for (Buf *b : bufs)
            if (b->p) (void)hipFree(b->p);
    for (Buf *b : bufs) {
        b->p = nullptr;
        b->cap = 0;
    }
    e->ext_ws = static_cast<float *>(dev_ptr);
    e->ext_ws_bytes = bytes;
    return OMNITOK_OK;
}

extern "C" int64_t omnitok_engine_workspace_bytes(omnitok_engine *e) {
    int64_t n = 0;
    for (Buf *b : {&e->X, &e->X2, &e->Y, &e->QKV, &e->AO, &e->HD, &e->Z, &e->ST}) n += b->cap * 4;
    return n;
}

extern "C" int omnitok_engine_set_option(omnitok_engine *e, const char *name, int value) {
    OT_CHECK_ARG(e && name, "engine_set_option: null pointer");
    OT_CHECK_ARG(value >= -1, "engine_set_option: value %d (-1 = follow the process default)", value);
    const std::string n(name);
    if (n == "gemm_mode") {
        OT_CHECK_ARG(value <= 2, "engine_set_option: gemm_mode %d", value);
        e->opt_gemm_mode = value;
    } else if (n == "attn_mode") {
        e->opt_attn_mode = value;
    } else if (n == "attn_vpack") {
        e->opt_attn_vpack = value;
    } else if (n == "gemm_pl") {
        e->opt_gemm_pl = value;
    } else {
        set_error("engine_set_option: %s is not a per-engine option (gemm_mode, attn_mode, attn_vpack, gemm_pl)", name);
        return OMNITOK_ERR_INVALID;
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_engine_set_timing(omnitok_engine *e, int enabled) {
    OT_CHECK_ARG(e, "set_timing: null engine");
    e->timing = enabled != 0;
    return OMNITOK_OK;
}

// returns the decode-side id range check of the last decode calls (1 = an id was out of range)
extern "C" int omnitok_engine_check_ids(omnitok_engine *e, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int flag = 0;
    OT_HIP(hipMemcpyAsync(&flag, e->err_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    OT_HIP(hipStreamSynchronize(stream));
    if (flag) {
        if (int rc = device_fill_u32(e->err_flag, 0u, 1, stream)) return rc;
        set_error("decode: token id out of range [0, %d)", e->cfg.n_codes);
        return OMNITOK_ERR_INVALID;
    }
    return OMNITOK_OK;
}

extern "C" int omnitok_engine_timing_report(omnitok_engine *e, char *buf, int buflen) {
    OT_CHECK_ARG(e && buf && buflen > 0, "timing_report: bad arguments");
    OT_HIP(hipDeviceSynchronize());
    struct Acc { int calls = 0; double ms = 0, work = 0; };
    std::map<std::string, Acc> acc;
    for (auto &r : e->recs) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        Acc &a = acc[r.name];
        a.calls++;
        a.ms += ms;
        a.work += r.work;
        e->pool.push_back(r.a);
        e->pool.push_back(r.b);
    }
    e->recs.clear();
    std::string s;
    char line[256];
    for (auto &kv : acc) {
        snprintf(line, sizeof(line), "%s %d %.6f %.6e\n", kv.first.c_str(), kv.second.calls, kv.second.ms,
                 kv.second.work);
        s += line;
    }
    strncpy(buf, s.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
    return OMNITOK_OK;
}
