// Internals of the engine (VQGAN.encode() / VQGAN.decode() as one native call each), shared by its three translation
// units:  engine_build.hip  the weight side: state_dict contract, repacked derivatives, static operand ranges
//         engine_run.hip    one Transformer: the block walk and its GEMM dispatch (reference attention.py:655-689)
//         engine.hip        shapes, workspace planner, the C ABI of encode / decode
#pragma once
#include "common.h"

#include <map>
#include <string>
#include <vector>
#include <algorithm>
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
    // plane data flow: [to_q o gamma | to_kv] (the LayerNorm gain folded into the Q rows; key of the gemm_pl packing),
    // b = to_q beta, u = row sums of to_kv: with the centred operand xc = x - mean,  LN(x) . Wq^T = rstd (xc . (Wq o gamma)^T) + b
    // and x . Wkv^T = xc . Wkv^T + mean u   (gemm_pl.h fold_*)
    const float *wqkv_fold = nullptr, *fold_b = nullptr, *fold_u = nullptr;
    const float *q_scale, *k_scale;
    float ln_bound = 0.0f;  // >= max |LN(x)|            (ranges for the fp16-split GEMM, gemm_h2.hip)
    float vnorm = 0.0f;     // max_j ||Wv_j||_2: |attention output| <= max_rows ||x||_2 * vnorm
    float q_amax = 0.0f, k_amax = 0.0f;  // max |q_scale|, max |k_scale|: ranges of the l2-normalised q / k (attn_h2.hip)
    std::string bias_prefix;  // spatial_rel_pos_bias prefix ("" if none)
    // temporal attention over ONE token (images): softmax over a single key is exactly 1, the block is x + (x Wv^T) Wo^T.
    // wvo = Wo . Wv (fp64 accumulation, rounded once) lets the plane data flow run it as ONE GEMM from the planes of x
    const float *wvo = nullptr;
    // fused temporal stage (gemm_pl.h PL_TSCORE): the folded to_q | to_k rows regrouped per head, [q_h (64) | k_h (64)] x heads,
    // with their fold vectors in the same order (key of the gemm_pl packing: wqk_t)
    const float *wqk_t = nullptr, *qk_t_fb = nullptr, *qk_t_fu = nullptr;
};
struct LayerW {  // 'w' block
    const float *ng, *nb, *wqkv, *wproj, *bproj, *bias_dense;
    const float *wqkv_fold = nullptr, *fold_b = nullptr, *fold_u = nullptr;  // all 3D rows folded (q, k, v from LN(x))
    float ln_bound = 0.0f, ao_bound = 0.0f;  // >= max |LN(x)|, >= max |window attention output|
    // >= max |q * scale|, >= max |k|: |LN(x) . W_j| <= ||LN(x)||_2 ||W_j||_2 (operand ranges of the packed window attention)
    float q_bound = 0.0f, k_bound = 0.0f;
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
    int opt_pl_min_tokens = -1, opt_temporal_chunk = -1;
    bool finalized = false;
    int inner_pad = 0;
    TransformerW enc_s, enc_t, dec_s, dec_t;
    // patch-embed / to_pixels GEMM operands, [0] first frame, [1] the rest (original tensors, or
    // K-padded / BatchNorm-folded / transposed copies made by finalize)
    const float *pe_w[2] = {nullptr, nullptr}, *pe_b[2] = {nullptr, nullptr};
    int pe_k[2] = {0, 0}, pe_ld[2] = {0, 0};
    const float *px_w[2] = {nullptr, nullptr}, *px_b[2] = {nullptr, nullptr};
    // derived
    float *cb_packed = nullptr, *cb_ee = nullptr, *cb_screen = nullptr, *alibi = nullptr;
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

constexpr int N_BOUND_LAUNCHES = 64;  // row-statistics launches with ranges per encode / decode

// process defaults of the per-engine modes (omnitok_set_option; engine_run.hip)
extern int g_gemm_mode, g_attn_mode, g_attn_vpack, g_gemm_pl, g_pl_min_tokens, g_temporal_chunk, g_prevq_fuse, g_vq_screen, g_temporal_fused;

Geo geometry(const omnitok_config &c);
bool walk_enc_grid(const char *block, int *gh, int *gw, int64_t *peak);
void add_transformer_spec(omnitok_engine *e, const std::string &prefix, const std::string &block, bool rel);
bool key_unused(const omnitok_engine *e, const std::string &k);
int alloc_f(omnitok_engine *e, float **out, int64_t n);
int ensure(Buf &b, int64_t n);
const char *k_embed(const omnitok_config &c);
const char *k_pre_w(const omnitok_config &c);
const char *k_pre_b(const omnitok_config &c);
const char *k_post_w(const omnitok_config &c);
const char *k_post_b(const omnitok_config &c);
const float *W(omnitok_engine *e, const std::string &k);
int build_transformer(omnitok_engine *e, TransformerW &tw, const std::string &prefix, const std::string &block, bool spatial,
                      hipStream_t stream);
int build_patch_operands(omnitok_engine *e, hipStream_t stream);
int get_rope(omnitok_engine *e, int N, const float **cosp, const float **sinp, hipStream_t stream);
int get_bias_table(omnitok_engine *e, const std::string &prefix, int gh, int gw, const float **out, hipStream_t stream);
int gemm_mode_of(const omnitok_engine *e);
int attn_mode_of(const omnitok_engine *e);
int attn_vpack_of(const omnitok_engine *e);
int gemm_pl_of(const omnitok_engine *e, int64_t tokens);
bool x3_ok(const omnitok_engine *e, int N, int K, int flags);
float *next_bounds(omnitok_engine *e);
struct VPack {  // packed-V output of the merged q|k|v launch (gemm_h2.hip): planes, first V column, sequence shape, |v| bound
    void *planes; int col0, n_tokens, heads; float bound; const float *bound_dev;
};
int eg_gemm(omnitok_engine *e, const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
            const float *residual, int64_t ldr, float *c, int64_t ldc, int64_t M, int N, int K, int flags,
            int64_t rpg, int64_t gstride, int64_t goff, hipStream_t stream, ABound ab = ABound(),
            const float *ln_stats = nullptr, const float *ln_g = nullptr, const float *ln_b = nullptr,
            int ln_cols = 0, float ln_bound = 0.0f, float *c2 = nullptr, int64_t ldc2 = 0, int split_col = 0,
            const VPack *vpk = nullptr, bool *vpacked = nullptr);
// pre_vq fused behind the last norm_out of the encoder (omnitok_layernorm_prevq): z [L, 8] instead of LayerNorm rows in X
struct PreVqFuse {
    const float *w, *b;
    float *z;
    int l2;
};
// out_planes: the final LayerNorm goes to Y as hi|lo operand planes (static scale of tw.out_bound) instead of fp32 rows in X
int run_transformer(omnitok_engine *e, const TransformerW &tw, int B, int T, int *ghp, int *gwp, bool spatial,
                    hipStream_t stream, bool transpose_out = false, bool out_planes = false, const PreVqFuse *pv = nullptr);
int reset_bounds(omnitok_engine *e, int B, hipStream_t stream);
void workspace_widths(const omnitok_engine *e, int64_t (&wd)[8]);
int64_t workspace_bytes_for(const omnitok_engine *e, int64_t L);
int ensure_workspace(omnitok_engine *e, int64_t L);

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

}  // namespace omnitok
