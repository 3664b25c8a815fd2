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
#include "engine.h"

namespace omnitok {

int reset_bounds(omnitok_engine *e, int B, hipStream_t stream) {
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
void workspace_widths(const omnitok_engine *e, int64_t (&wd)[8]) {
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

int64_t workspace_bytes_for(const omnitok_engine *e, int64_t L) {
    int64_t wd[8], total = 0;
    workspace_widths(e, wd);
    L = (L + 255) / 256 * 256;  // plane operands are read in whole 256-row tiles (gemm_pl.h)
    for (int i = 0; i < 8; ++i) total += ((L * wd[i] * 4 + 255) / 256) * 256;
    return total;
}

int ensure_workspace(omnitok_engine *e, int64_t L) {
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
// fuse_prevq: when nothing sits between the temporal Transformer's norm_out and pre_vq (no deferred pools) the two run as
// one pass and the latent z lands in e->Z (*prevq_done = true); X then does NOT hold the normalised tokens.
static int encode_tokens(omnitok_engine *e, const float *x, int B, int F, int H, int W_, int *T_out, int *S_out,
                         hipStream_t stream, bool fuse_prevq = false, bool *prevq_done = nullptr) {
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
    const bool fuse = fuse_prevq && g_prevq_fuse && !g.defer_s && !(g.defer_t && T > 1) && (D == 256 || D == 512 || D == 1024) &&
                      c.codebook_dim == 8;
    const PreVqFuse pvf{fuse ? W(e, k_pre_w(c)) : nullptr, fuse ? W(e, k_pre_b(c)) : nullptr, e->Z.p, c.l2_code};
    if (int rc = run_transformer(e, e->enc_t, B, T, &gh, &gw, false, stream, true, false, fuse ? &pvf : nullptr)) return rc;
    if (prevq_done) *prevq_done = fuse;
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
    bool prevq_done = false;
    if (int rc = encode_tokens(e, x, B, F, H, W_, &T, &S, stream, true, &prevq_done)) return rc;
    const omnitok_config &c = e->cfg;
    const int D = c.dim;
    const int64_t L = (int64_t)B * T * S;
    // ---- pre_vq + l2norm + nearest code (reference omnitokenizer.py:248-255) ----------------
    if (!prevq_done)
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
        if (g_vq_screen && e->cb_screen)  // same ids, bit for bit (csrc/vq.hip vq_screen_kernel)
            OT_RUN("vq_argmin", 2.0 * (double)L * c.n_codes * 8.0,
                   omnitok_vq_argmin_screened(e->Z.p, e->cb_packed, e->cb_ee, e->cb_screen, L, c.n_codes, ids_out, stream));
        else
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
    // Plane data flow of to_pixels: the last Transformer's norm_out writes LayerNorm(x) as operand planes (Y) instead of fp32
    // rows, and the two pixel GEMMs scatter their fp32 results straight into the video (un-patchify epilogue): no token
    // tensor after norm_out, no [L, K] pixel rows, no un-patchify pass.
    bool plain_blocks = true;  // no pooling / Up blocks in the last Transformer: its grid is the token grid
    for (const char *q = c.dec_block; *q; ++q) plain_blocks = plain_blocks && (*q == 't' || *q == 'w');
    const bool px_pl = plain_blocks && gemm_pl_of(e, (int64_t)B * T2 * gh2 * gw2) && gemm_mode_of(e) == 2 && D == 512 && !c.patch_embed_cnn && S % 256 == 0 && p == 8 && gw2 % 32 == 0 &&
                       aligned16(pixels_out) &&  // the un-patchify epilogue's own preconditions (gemm_pl.hip PL_UNPATCH): a caller
                                                 // whose output is only 4-byte aligned takes the gemm + unpatchify path instead
                       e->plw.count(e->px_w[0]) && (T2 == 1 || e->plw.count(e->px_w[1])) && e->dec_s.out_bound > 0.0f &&
                       e->dec_s.out_bound < 1e30f;
    if (int rc = run_transformer(e, e->dec_s, B, T2, &ghc, &gwc, true, stream, false, px_pl)) return rc;
    if (px_pl) {
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && T2 == 1) break;
            const int tt = i ? T2 - 1 : 1, Kp = i ? K1 : K0;
            const H2W &wp = e->plw[e->px_w[i]];
            omnitok_pl_gemm g{};
            g.a = e->Y.p;
            g.a_scale_const = omnitok_pl_unscale(e->dec_s.out_bound);
            g.a_rpg = (int64_t)tt * S;
            g.a_gstride = (int64_t)T2 * S;
            g.a_goff = i ? S : 0;
            g.w = wp.pl;
            g.w_scale = wp.sc;
            g.bias = e->px_b[i];
            g.c = pixels_out;
            g.epilogue = 5;
            g.up_C = C; g.up_F = F; g.up_H = H; g.up_W = W_; g.up_f0 = i; g.up_t = tt; g.up_pt = i ? pt : 1; g.up_p = p;
            g.M = (int64_t)B * tt * S;
            g.N = Kp;
            g.K = D;
            OT_RUN("gemm_pixels", 2.0 * g.M * (double)Kp * D, omnitok_gemm_pl(&g, stream));
        }
        return OMNITOK_OK;
    }
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
    } else if (n == "pl_min_tokens") {
        e->opt_pl_min_tokens = value;
    } else if (n == "temporal_chunk") {
        e->opt_temporal_chunk = value;
    } else {
        set_error("engine_set_option: %s is not a per-engine option (gemm_mode, attn_mode, attn_vpack, gemm_pl, pl_min_tokens, "
                  "temporal_chunk)", name);
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

