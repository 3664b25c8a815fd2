// Engine, one Transformer: the block walk (PEG, attention / window attention, pooling / Up blocks, FeedForward; residual
// order and grid tracking of reference attention.py:655-689) and the dispatch of its nn.Linear layers onto the GEMM kernels
// of the engine's mode (plane data flow gemm_pl.h | fp16 split gemm_h2.hip | bf16 split gemm_x3.hip | fp32 MFMA gemm.hip).
#include "engine.h"

namespace omnitok {

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
// "pl_min_tokens" (default 0 since r06: ONE data flow at every call size): calls with fewer tokens (B * T' * h * w) than this
// take the gemm_pl 0 data flow even when gemm_pl is 1.  Rounds 4-5 shipped 12288 here because the plane GEMM only had 256 x 256
// tiles (one 256^2 image 4.09 ms vs 2.00 ms, profiles/r04_small_batch_latency.txt) -- and a clip's low bits then depended on
// the size of the call it was in.  With the thin-tile family of gemm_pl.hip (128 x 128 / 128 x 64 / 32- and 64-row LayerNorm
// tiles, same bits as the big tiles) the plane flow is as fast or faster at every size (one image 2.40 vs 2.39 ms of kernel
// time, one clip 4.23 vs 4.44, two clips 5.86 vs 6.39: profiles/r06_small_calls.txt), so a clip is now encoded and decoded to
// the same bits alone and inside any batch.  The option stays as an A/B switch (also per engine:
// omnitok_engine_set_option "pl_min_tokens", -1 = this process default).
int g_gemm_pl = 1;
int g_pl_min_tokens = 0;
// "temporal_chunk" (clips per chunk, 0 = off): a temporal 't' block runs its q|k|v plane GEMM and the temporal attention
// kernel chunk by chunk -- GEMM(chunk) -> attention(chunk) back to back through ONE chunk-sized fp32 q|k|v buffer (31.5 MB
// per 17x256^2 clip), so that the attention kernel's reads hit the 256 MiB Infinity Cache instead of HBM
// (profiles/r05_temporal_chunk.txt).  Results are bit-identical to the unchunked launch: tiles never straddle a clip.
int g_temporal_chunk = 0;
// "prevq_fuse" 1 (default): omnitok_encode runs pre_vq inside the encoder's last LayerNorm pass (omnitok_layernorm_prevq,
// bit-identical) whenever nothing sits between the two (no deferred pools); 0: LayerNorm store, then omnitok_pre_vq.
int g_prevq_fuse = 1;
// "temporal_fused" 1 (default): temporal 't' blocks with T' == 5 (17-frame clips at temporal_patch_size 4), causal, on the plane
// data flow run WITHOUT the fp32 q|k|v round trip (2 GB per block at C3): omnitok_stats_pack_temporal permutes the rows so that
// a sequence's five steps sit in one lane pair, a q|k GEMM forms the softmax weights in its epilogue (PL_TSCORE: the k wave hands
// k_s to the q wave through LDS) and a V GEMM applies them in its own (PL_TPV), writing the out-projection's operand planes.
// C3: 0.616 + 0.339 ms per block against 0.658 + 0.263 for the q|k|v GEMM + attn_temporal pair, but 2 GB less HBM traffic per
// block on a power-capped step: -0.86 ms per step (profiles/r05_temporal_fused.txt).  0: q|k|v GEMM + attn_temporal kernel
// (what every other T', non-causal configurations and grids with S % 64 != 0 take anyway).
int g_temporal_fused = 1;
// "qkv_pl" 1 (default, with gemm_pl): the q|k|v projection as a plane GEMM too -- one pass writes the row statistics and
// the centred rows x - mean as planes (omnitok_stats_pack), the LayerNorm of the Q columns is folded into the weight and finished in the
// epilogue, which for spatial attention also does RoPE + l2norm + scales and writes Q and K straight into the attention
// kernel's packed planes (no fp32 Q / K, no attn_pack pass); V by a swapped-orientation launch.  0: gemm_h2 with the
// LayerNorm fused into its A loader (r02 form).
int g_qkv_pl = 1;
// "attn_window_mode" 1 (default, with gemm_pl / qkv_pl): window attention on the fp16 matrix cores from packed operands --
// the q|k|v plane GEMM runs on window-major rows (omnitok_stats_pack_windows) and its packing epilogues write what
// omnitok_attn_window_h2 reads; no fp32 q|k|v in memory.  0: fp32 q|k|v + the fp32-MFMA kernel of attn_spatial.hip.
int g_attn_window_mode = 1;

int gemm_mode_of(const omnitok_engine *e) { return e->opt_gemm_mode >= 0 ? e->opt_gemm_mode : g_gemm_mode; }
int attn_mode_of(const omnitok_engine *e) { return e->opt_attn_mode >= 0 ? e->opt_attn_mode : g_attn_mode; }
int attn_vpack_of(const omnitok_engine *e) { return e->opt_attn_vpack >= 0 ? e->opt_attn_vpack : g_attn_vpack; }
int gemm_pl_of(const omnitok_engine *e, int64_t tokens) {
    const int min_tokens = e->opt_pl_min_tokens >= 0 ? e->opt_pl_min_tokens : g_pl_min_tokens;
    return (e->opt_gemm_pl >= 0 ? e->opt_gemm_pl : g_gemm_pl) && tokens >= min_tokens;
}
static int temporal_chunk_of(const omnitok_engine *e) { return e->opt_temporal_chunk >= 0 ? e->opt_temporal_chunk : g_temporal_chunk; }

bool x3_ok(const omnitok_engine *e, int N, int K, int flags) {
    return gemm_mode_of(e) >= 1 && N % 32 == 0 && K % 32 == 0 && !(flags & OMNITOK_GEMM_LEAKY) &&
           (!(flags & OMNITOK_GEMM_GEGLU) || N % 64 == 0);
}

float *next_bounds(omnitok_engine *e) {  // [n_clips][2] slots of one row-statistics launch
    if (e->bound_next >= N_BOUND_LAUNCHES) return nullptr;
    return e->bounds + (int64_t)2 * e->bound_clips * (e->bound_next++);
}



int eg_gemm(omnitok_engine *e, const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                   const float *residual, int64_t ldr, float *c, int64_t ldc, int64_t M, int N, int K, int flags,
                   int64_t rpg, int64_t gstride, int64_t goff, hipStream_t stream, ABound ab,
                   const float *ln_stats, const float *ln_g, const float *ln_b,
                   int ln_cols, float ln_bound, float *c2, int64_t ldc2, int split_col,
                   const VPack *vpk, bool *vpacked) {
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
int run_transformer(omnitok_engine *e, const TransformerW &tw, int B, int T, int *ghp, int *gwp, bool spatial,
                           hipStream_t stream, bool transpose_out, bool out_planes, const PreVqFuse *pv) {
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
    bool pl_ok = gemm_pl_of(e, L) && gemm_mode_of(e) == 2 && fused && D == 512 && x3_ok(e, 2 * e->inner_pad, D, OMNITOK_GEMM_GEGLU);
    // the plane producers scale by constants derived from the static operand bounds: a degenerate FeedForward LayerNorm
    // (gamma == beta == 0 gives bound 0) or a non-finite bound sends the whole Transformer down the gemm_h2 / x3 branch,
    // which handles missing ranges by itself
    for (const Layer &ly : tw.layers)
        if (!(ly.ff.ln_bound > 0.0f && ly.ff.ln_bound < 1e30f && ly.ff.h_bound > 0.0f && ly.ff.h_bound < 1e30f)) pl_ok = false;
    const bool pl = pl_ok;
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
                if (pl && ly.t.wvo && e->plw.count(ly.t.wvo)) {
                    // plane data flow: x as planes (one power-of-two scale per row) -> ONE GEMM on Wo . Wv with the residual
                    // add and the FeedForward's LayerNorm planes in its epilogue.  (x Wv^T) Wo^T = x (Wo Wv)^T in exact
                    // arithmetic; the composed weight is rounded once from fp64, the two-step form rounds V to fp32 in
                    // between -- both inside the path's fp32 noise (golden image fixtures, tests/test_gpu_e2e.py).
                    const int64_t Lp = (L + 255) / 256 * 256;
                    OT_RUN("stats_pack", 2.0 * L * D * 4.0,
                           omnitok_stats_pack(e->X.p, L, D, 1e-5f, 0, e->X2.p, Lp, e->Z.p, nullptr, nullptr, 0, stream));
                    if (int rc = gemm_out_pl(ly, e->X2.p, e->Z.p, 1.0f, ly.t.wvo, nullptr)) return rc;
                    goto feed_forward;
                }
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
            if (!spatial && T == 5 && pl && g_qkv_pl && g_temporal_fused && bs && c.causal_temporal && ly.t.wqk_t &&
                e->plw.count(ly.t.wqk_t) && ly.t.wqkv_fold && e->plw.count(ly.t.wqkv_fold) && S % 64 == 0 && heads * 64 == D &&
                ly.t.vnorm > 0.0f) {
                // fused temporal stage (gemm_pl.h PL_TSCORE / PL_TPV; reference attention.py:402-486 with is_spatial = False)
                const int64_t nseq = (int64_t)B * S;
                float *P = e->QKV.p, *out_scale = e->QKV.p + nseq * heads * 40;
                OT_RUN("stats_pack", 2.0 * L * D * 4.0,
                       omnitok_stats_pack_temporal(e->X.p, nseq, D, 1e-5f, e->X2.p, e->Z.p, e->ST.p, bs, S, stream));
                const float *alibi = c.legacy_attention ? e->alibi : nullptr;
                omnitok_pl_gemm g{};
                g.a = e->X2.p;
                g.a_scale = e->Z.p;
                g.fold_stats = e->ST.p;
                g.M = L;
                g.K = D;
                g.tp = P;
                g.t_nseq = (int)nseq;
                g.t_heads = heads;
                g.t_alibi = alibi;
                omnitok_pl_gemm q = g;
                const H2W &wqk = e->plw[ly.t.wqk_t];
                q.w = wqk.pl;
                q.w_scale = wqk.sc;
                q.fold_b = ly.t.qk_t_fb;
                q.fold_u = ly.t.qk_t_fu;
                q.N = 2 * D;
                q.epilogue = 6;
                q.q_scale = ly.t.q_scale;
                q.k_scale = ly.t.k_scale;
                q.q_mul = 8.0f;
                OT_RUN("gemm_qkv", gemm_f * 2 * D, omnitok_gemm_pl(&q, stream));
                const H2W &wf = e->plw[ly.t.wqkv_fold];
                omnitok_pl_gemm v = g;
                v.w = static_cast<const char *>(wf.pl) + (int64_t)(2 * D / 64) * (D / 32) * 8192;
                v.w_scale = wf.sc + 2 * D;
                v.fold_u = ly.t.fold_u + 2 * D;
                v.N = D;
                v.epilogue = 7;
                v.out_planes = e->AO.p;
                v.out_planes_k = D;
                v.t_out_scale = out_scale;
                v.v_bound = ab_ao.stat;
                v.v_bound_dev = ab_ao.dev;
                v.v_bound_stride = 2;
                v.t_seqs_per_clip = S;
                OT_RUN("gemm_qkv", gemm_f * D, omnitok_gemm_pl(&v, stream));
                if (int rc = gemm_out_pl(ly, e->AO.p, out_scale, 1.0f, ly.t.wo, nullptr)) return rc;
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
            bool qk_packed = false;  // the q|k launch wrote the packed Q / K planes itself (no attn_pack pass)
            bool temporal_done = false;  // the chunked temporal stage ran GEMM and attention already
            const bool qkv_pl = pl && g_qkv_pl && ly.t.wqkv_fold && e->plw.count(ly.t.wqkv_fold) && bs;
            if (qkv_pl) {
                const int64_t Lp = (L + 255) / 256 * 256;
                float *a_sc = e->Z.p;  // [L] row scales of the raw planes (Z is free inside a Transformer)
                OT_RUN("stats_pack", 2.0 * L * D * 4.0,
                       omnitok_stats_pack(e->X.p, L, D, 1e-5f, 1, e->X2.p, Lp, a_sc, e->ST.p, bs, rpc, stream));
                const H2W &wf = e->plw[ly.t.wqkv_fold];
                omnitok_pl_gemm g{};
                g.a = e->X2.p;
                g.a_scale = a_sc;
                g.w = wf.pl;
                g.w_scale = wf.sc;
                g.fold_stats = e->ST.p;
                g.fold_b = ly.t.fold_b;
                g.fold_u = ly.t.fold_u;
                g.fold_cols = D;
                g.M = L;
                g.K = D;
                if (attn_h2 && rpc % 256 == 0) {
                    // spatial attention on packed operands: Q | K through the packing epilogue, V through the swapped launch
                    const float *cosp = nullptr, *sinp = nullptr;
                    if (c.spatial_rope)
                        if (int rc = get_rope(e, S, &cosp, &sinp, stream)) return rc;
                    omnitok_pl_gemm q = g;
                    q.N = 2 * D;
                    q.epilogue = 4;
                    q.qp = qp;
                    q.kp = kp;
                    q.qk_k0 = D;
                    q.n_tokens = S;
                    q.heads = heads;
                    q.rope_cos = cosp;
                    q.rope_sin = sinp;
                    q.q_scale = ly.t.q_scale;
                    q.k_scale = ly.t.k_scale;
                    q.q_mul = 8.0f;
                    q.q_bound = qb;
                    q.k_bound = kb;
                    OT_RUN("gemm_qkv", gemm_f * 2 * D, omnitok_gemm_pl(&q, stream));
                    omnitok_pl_gemm v = g;
                    v.fold_cols = 0;                 // every V column: xc . Wv^T + mean u
                    v.fold_b = nullptr;
                    v.fold_u = ly.t.fold_u + 2 * D;
                    v.w = static_cast<const char *>(wf.pl) + (int64_t)(2 * D / 64) * (D / 32) * 8192;
                    v.w_scale = wf.sc + 2 * D;
                    v.N = D;
                    v.epilogue = 3;
                    v.vp = vp;
                    v.n_tokens = S;
                    v.heads = heads;
                    v.v_bound = ab_ao.stat;
                    v.v_bound_dev = ab_ao.dev;
                    v.v_bound_stride = 2;
                    v.rows_per_clip = rpc;
                    OT_RUN("gemm_qkv", gemm_f * D, omnitok_gemm_pl(&v, stream));
                    vpacked = true;
                    qk_packed = true;
                } else {
                    g.N = 3 * D;
                    g.epilogue = 0;
                    g.c = Q;
                    g.ldc = D;
                    g.c2 = KV;
                    g.ldc2 = 2 * D;
                    g.c_split_n = D;
                    const int chunk = (!spatial && T > 1) ? temporal_chunk_of(e) : 0;
                    if (chunk > 0 && chunk < B && T <= 17 && S % 16 == 0 && rpc % 256 == 0) {
                        // "temporal_chunk": GEMM(chunk) -> attention(chunk) through one chunk-sized q|k|v buffer that stays
                        // in the Infinity Cache between the two kernels.  A clip is a whole number of 256-row tiles and of
                        // 64-row plane blocks, so every launch computes exactly what the full launch computes for its rows.
                        const float *alibi = (c.legacy_attention && c.causal_temporal) ? e->alibi : nullptr;
                        float *Qc = e->QKV.p, *KVc = e->QKV.p + (int64_t)chunk * rpc * D;
                        for (int c0 = 0; c0 < B; c0 += chunk) {
                            const int nb = std::min(chunk, B - c0);
                            const int64_t r0 = (int64_t)c0 * rpc, Lc = (int64_t)nb * rpc;
                            const int64_t pl_off = (r0 >> 6) * (int64_t)(D >> 5) * 8192;  // bytes: 8 KiB per 64 rows x 32 k
                            omnitok_pl_gemm gc = g;
                            gc.a = static_cast<const char *>(g.a) + pl_off;
                            gc.a_scale = a_sc + r0;
                            gc.fold_stats = e->ST.p + 2 * r0;
                            gc.c = Qc;
                            gc.c2 = KVc;
                            gc.M = Lc;
                            OT_RUN("gemm_qkv", 2.0 * (double)Lc * D * 3 * D, omnitok_gemm_pl(&gc, stream));
                            // out_scale rows [r0, r0 + Lc) of ST: behind every (mean, rstd) pair a later chunk still reads
                            OT_RUN("attn_temporal", 4.0 * Lc * D * 4.0,
                                   omnitok_attn_temporal_planes(Qc, D, KVc, KVc + D, 2 * D, nullptr, D,
                                                                reinterpret_cast<char *>(e->AO.p) + pl_off, e->ST.p + r0, ab_ao.stat,
                                                                ab_ao.dev + 2 * c0, 2, S, (int64_t)nb * S, T, heads, ly.t.q_scale,
                                                                ly.t.k_scale, 8.0f, c.causal_temporal, alibi, stream));
                        }
                        temporal_done = true;
                        ao_planes = true;
                    } else {
                        OT_RUN("gemm_qkv", gemm_f * 3 * D, omnitok_gemm_pl(&g, stream));
                    }
                }
            } else if (fused && D % 256 == 0) {
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
                    if (!qk_packed)
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
            } else if (!temporal_done) {
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
            const auto okb = [](float v) { return v > 0.0f && v < 1e30f; };
            if (pl && g_qkv_pl && g_attn_window_mode == 1 && ly.w.wqkv_fold && e->plw.count(ly.w.wqkv_fold) && c.window_size == 8 &&
                gh % 8 == 0 && gw % 8 == 0 && c.dim_head == 64 && okb(ly.w.q_bound) && okb(ly.w.k_bound) && okb(ly.w.ao_bound)) {
                // Window attention on packed operands (reference attention.py:254-293): the centred rows go out in
                // window-major order, so that every 64 consecutive GEMM rows are one window and the packing epilogues (n_tokens
                // = 64; no RoPE, no l2norm, q * head scale) write the attention kernel's Q / K / V blocks directly; the
                // attention kernel maps its output rows back to token order (window_reverse) as planes for the proj GEMM.
                const int64_t Lp = (L + 255) / 256 * 256;
                OT_RUN("stats_pack", 2.0 * L * D * 4.0,
                       omnitok_stats_pack_windows(e->X.p, L, D, 1e-5f, 1, e->X2.p, Lp, e->Z.p, e->ST.p, gh, gw, 8, stream));
                const H2W &wf = e->plw[ly.w.wqkv_fold];
                unsigned char *qp = reinterpret_cast<unsigned char *>(e->Y.p);
                unsigned char *kp = reinterpret_cast<unsigned char *>(e->HD.p);
                unsigned char *vp = kp + (size_t)L * D * 4;
                omnitok_pl_gemm g{};
                g.a = e->X2.p;
                g.a_scale = e->Z.p;
                g.w = wf.pl;
                g.w_scale = wf.sc;
                g.fold_stats = e->ST.p;
                g.fold_b = ly.w.fold_b;
                g.M = L;
                g.K = D;
                omnitok_pl_gemm q = g;
                q.N = 2 * D;
                q.fold_cols = 2 * D;
                q.epilogue = 4;
                q.qp = qp;
                q.kp = kp;
                q.qk_k0 = D;
                q.n_tokens = 64;
                q.heads = heads;
                q.q_mul = 1.0f / sqrtf((float)c.dim_head);
                q.q_bound = ly.w.q_bound;
                q.k_bound = ly.w.k_bound;
                OT_RUN("gemm_qkv", gemm_f * 2 * D, omnitok_gemm_pl(&q, stream));
                omnitok_pl_gemm v = g;
                v.w = static_cast<const char *>(wf.pl) + (int64_t)(2 * D / 64) * (D / 32) * 8192;
                v.w_scale = wf.sc + 2 * D;
                v.fold_b = ly.w.fold_b + 2 * D;
                v.fold_cols = D;
                v.N = D;
                v.epilogue = 3;
                v.vp = vp;
                v.n_tokens = 64;
                v.heads = heads;
                v.v_bound = ly.w.ao_bound;
                OT_RUN("gemm_qkv", gemm_f * D, omnitok_gemm_pl(&v, stream));
                OT_RUN("attn_window", 4.0 * (double)L * 64.0 * D,
                       omnitok_attn_window_h2(qp, kp, vp, ly.w.bias_dense, nullptr, 0, e->AO.p, ly.w.q_bound, ly.w.k_bound,
                                              ly.w.ao_bound, B * T, gh, gw, heads, stream));
                if (int rc = gemm_out_pl(ly, e->AO.p, nullptr, omnitok_pl_unscale(ly.w.ao_bound), ly.w.wproj, ly.w.bproj))
                    return rc;
                goto feed_forward;
            }
            if (pl && g_qkv_pl && ly.w.wqkv_fold && e->plw.count(ly.w.wqkv_fold)) {
                // q, k, v all from LN(x) (reference attention.py:262-272): the gain folded into every weight row
                const int64_t Lp = (L + 255) / 256 * 256;
                OT_RUN("stats_pack", 2.0 * L * D * 4.0,
                       omnitok_stats_pack(e->X.p, L, D, 1e-5f, 1, e->X2.p, Lp, e->Z.p, e->ST.p, nullptr, 0, stream));
                const H2W &wf = e->plw[ly.w.wqkv_fold];
                omnitok_pl_gemm g{};
                g.a = e->X2.p;
                g.a_scale = e->Z.p;
                g.w = wf.pl;
                g.w_scale = wf.sc;
                g.fold_stats = e->ST.p;
                g.fold_b = ly.w.fold_b;
                g.fold_cols = 3 * D;
                g.c = e->QKV.p;
                g.ldc = 3 * D;
                g.M = L;
                g.N = 3 * D;
                g.K = D;
                OT_RUN("gemm_qkv", gemm_f * 3 * D, omnitok_gemm_pl(&g, stream));
            } else if (fused) {
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
            h.k_valid = c.ff_inner;  // the hidden's pad columns are exactly 0 (GEGLU epilogue) and so are the weight's
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
    if (out_planes) {  // the consumer is a plane GEMM (to_pixels): LayerNorm + operand split in one pass, rows stay in place
        OT_RUN("layernorm", 2.0 * L * D * 4.0,
               omnitok_layernorm_planes(e->X.p, L, D, 1e-5f, tw.og, tw.ob, tw.out_bound, e->Y.p, (L + 255) / 256 * 256, stream));
        *ghp = gh;
        *gwp = gw;
        return OMNITOK_OK;
    }
    if (pv) {  // encoder end: LayerNorm -> pre_vq -> l2norm in one pass, z rows in the order the LayerNorm would have stored
        OT_RUN("pre_vq", 1.0 * L * D * 4.0,
               omnitok_layernorm_prevq(e->X.p, tw.og, tw.ob, pv->w, pv->b, pv->z, B, spatial ? T : S, spatial ? S : T, D, 1e-5f,
                                       transpose_out && T > 1, pv->l2, stream));
        *ghp = gh;
        *gwp = gw;
        return OMNITOK_OK;
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

}  // namespace omnitok
