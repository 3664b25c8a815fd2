"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the OmniTokenizer encode/decode path.

A functional fp32 restatement (plain torch CPU ops on a flat state_dict, no nn.Module, no
einops) of the algorithm in the reference's three hot-path files.  Every function cites the
reference lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; the product (omnitokenizer_amd/) never does and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the reference itself, run in the build container through
oracle/ref_harness.py: tests/golden/*.npz (made by tests/golden/make_golden.py) hold the
reference's ids / pixels / pre-VQ z for seeded weights+inputs, and
tests/test_oracle_vs_golden.py + tests/test_oracle_vs_reference.py check this file against them.

All arithmetic is fp32 (ids int64) like the reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------------


def layer_norm(x, weight, bias, eps=1e-5):
    """reference attention.py:73-80 (custom LayerNorm, beta = zero buffer) and nn.LayerNorm in
    FeedForward / patch-embed (attention.py:163, omnitokenizer.py:809-821); eps 1e-5."""
    return F.layer_norm(x, x.shape[-1:], weight, bias, eps)


def l2norm(t):
    """reference attention.py:24-25: F.normalize(t, dim=-1) = t / max(||t||_2, 1e-12)."""
    return F.normalize(t, dim=-1)


def rope_table(n_tokens: int, dim_head: int = 64, theta: float = 10000.0):
    """reference attention.py:28-43 precompute_freqs_cis_2d: H=int(sqrt(N)); x=pos%H, y=pos//H;
    freqs_i = theta^(-4i/dim) for i < dim/4; per token dim/2 angles interleaved (x0,y0,x1,y1,..).
    Returns (cos, sin) each [N, dim/2] fp32, computed with the same fp32 torch ops."""
    H = int(n_tokens ** 0.5)
    pos = torch.arange(0, n_tokens)
    x_pos, y_pos = pos % H, pos // H
    freqs = 1.0 / (theta ** (torch.arange(0, dim_head, 4)[: (dim_head // 4)].float() / dim_head))
    x_f = torch.outer(x_pos, freqs).float()
    y_f = torch.outer(y_pos, freqs).float()
    ang = torch.stack([x_f, y_f], dim=-1).reshape(n_tokens, -1)  # N, dim/2
    cis = torch.polar(torch.ones_like(ang), ang)
    return cis.real.contiguous(), cis.imag.contiguous()


def apply_rope(t, cos, sin):
    """reference attention.py:57-70 apply_rotary_emb on [B,N,H,D]: channel pairs (2j,2j+1) form
    complex numbers multiplied by cis[n,j]; the same table for every head."""
    tr = t.reshape(*t.shape[:-1], -1, 2)
    a, b = tr[..., 0], tr[..., 1]
    c = cos[None, :, None, :]
    s = sin[None, :, None, :]
    return torch.stack([a * c - b * s, a * s + b * c], dim=-1).flatten(3)


def alibi_slopes(heads: int):
    """reference attention.py:506-517 AlibiPositionalBias._get_slopes."""
    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(heads).is_integer():
        return pow2(heads)
    c = 2 ** math.floor(math.log2(heads))
    return pow2(c) + pow2(2 * c)[0::2][: heads - c]


def continuous_position_bias(sd, prefix, h, w):
    """reference attention.py:535-583 ContinuousPositionBias.forward(h, w) -> [heads, h*w, h*w].
    MLP(2->dim->dim->heads, LeakyReLU 0.1) on sign(d)*log(1+|d|) of every token pair."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack([ys, xs]).reshape(2, -1).t()  # (h w) 2
    rel = grid[:, None, :] - grid[None, :, :]
    rel = torch.sign(rel) * torch.log(rel.abs() + 1)
    r = rel.float().to(sd[f"{prefix}.net.0.0.weight"].dtype)  # fp32 like the reference (fp64 only for noise measurements)
    r = F.leaky_relu(F.linear(r, sd[f"{prefix}.net.0.0.weight"], sd[f"{prefix}.net.0.0.bias"]), 0.1)
    r = F.leaky_relu(F.linear(r, sd[f"{prefix}.net.1.0.weight"], sd[f"{prefix}.net.1.0.bias"]), 0.1)
    r = F.linear(r, sd[f"{prefix}.net.2.weight"], sd[f"{prefix}.net.2.bias"])
    return r.permute(2, 0, 1).contiguous()


def continuous_position_bias_table(sd, prefix, h, w):
    """The same MLP evaluated once per distinct offset: table[head, dy+h-1, dx+w-1].  The bias of
    pair (i,j) depends only on (row_i-row_j, col_i-col_j) (attention.py:567-574), so
    bias[hd,i,j] == table[hd, dy+h-1, dx+w-1]; checked in tests."""
    dy, dx = torch.meshgrid(torch.arange(-(h - 1), h), torch.arange(-(w - 1), w), indexing="ij")
    rel = torch.stack([dy, dx], dim=-1)
    rel = torch.sign(rel) * torch.log(rel.abs() + 1)
    r = rel.float().to(sd[f"{prefix}.net.0.0.weight"].dtype)  # fp32 like the reference (fp64 only for noise measurements)
    r = F.leaky_relu(F.linear(r, sd[f"{prefix}.net.0.0.weight"], sd[f"{prefix}.net.0.0.bias"]), 0.1)
    r = F.leaky_relu(F.linear(r, sd[f"{prefix}.net.1.0.weight"], sd[f"{prefix}.net.1.0.bias"]), 0.1)
    r = F.linear(r, sd[f"{prefix}.net.2.weight"], sd[f"{prefix}.net.2.bias"])
    return r.permute(2, 0, 1).contiguous()  # heads, 2h-1, 2w-1


# --------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------


def peg(x, weight, bias, shape, causal):
    """reference attention.py:298-338 PEG.forward.  x [Bn, N, D] is reshaped -- as a raw
    contiguous buffer -- to [B,T,H,W,D] (line 319) whatever token order it was written in
    (SURVEY.md A.1-Q5), zero-padded W(1,1) H(1,1) T(2,0 causal | 1,1), depthwise 3x3x3 conv
    (+bias), and viewed back.  The caller adds the residual (attention.py:667)."""
    B, T, H, W = shape
    D = x.shape[-1]
    y = x.reshape(B, T, H, W, D).permute(0, 4, 1, 2, 3)
    fp = (2, 0) if causal else (1, 1)
    y = F.pad(y, (1, 1, 1, 1, *fp), value=0.0)
    y = F.conv3d(y, weight, bias, groups=D)
    y = y.permute(0, 2, 3, 4, 1)
    return y.reshape(x.shape)


def attention(sd, p, x, cfg, is_spatial, causal, spatial_pos):
    """reference attention.py:402-486 Attention.forward (no context, no mask, no null kv).
    Q from LN(x), K/V from the raw x (kv_input captured at :407 before the norm at :409)."""
    Bn, N, _ = x.shape
    h, d = cfg.heads, cfg.dim_head
    xn = layer_norm(x, sd[f"{p}.norm.gamma"], sd[f"{p}.norm.beta"])
    q = F.linear(xn, sd[f"{p}.to_q.weight"])
    k, v = F.linear(x, sd[f"{p}.to_kv.weight"]).chunk(2, dim=-1)
    q, k, v = (t.reshape(Bn, N, h, d) for t in (q, k, v))
    if spatial_pos == "rope" and is_spatial:  # :417-421
        cos, sin = rope_table(N, d)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))  # b h n d
    q, k = l2norm(q), l2norm(k)  # :435
    q = q * sd[f"{p}.q_scale"]
    k = k * sd[f"{p}.k_scale"]
    scale = 8.0  # Attention(scale=8) default, attention.py:352
    if cfg.attention_mode == "sdpa":
        # :451 -- attn_bias / ALiBi are computed but never passed
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0,
                                             is_causal=causal, scale=scale)
    else:
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * scale  # :454
        if spatial_pos == "rel" and is_spatial:  # :456-466
            hh = ww = int(math.sqrt(N))
            sim = sim + continuous_position_bias(sd, f"{p}.spatial_rel_pos_bias", hh, ww)
        if causal:  # :473-478
            i = j = N
            ar = torch.arange(j)
            bias = -torch.abs(ar[None, None, :] - ar[None, :, None]).float()
            slopes = torch.tensor(alibi_slopes(h)).reshape(h, 1, 1)
            sim = sim + bias * slopes
            mask = torch.ones((i, j), dtype=torch.bool).triu(j - i + 1)
            sim = sim.masked_fill(mask, -torch.finfo(sim.dtype).max)
        out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(Bn, N, h * d)
    return F.linear(out, sd[f"{p}.to_out.weight"])


def window_attention(sd, p, x, cfg):
    """reference attention.py:254-293 WindowAttention.forward: LN, 8x8 (window_size) partition of
    the sqrt(N) x sqrt(N) grid, qkv (no bias), q*head_dim^-0.5, + relative_position_bias_table
    gathered by relative_position_index, softmax, PV, proj (+bias), window reverse."""
    B_, N, C = x.shape
    H = W = int(math.sqrt(N))
    ws, nh = cfg.window_size, cfg.heads
    xn = layer_norm(x, sd[f"{p}.norm.gamma"], sd[f"{p}.norm.beta"]).view(B_, H, W, C)
    xw = xn.view(B_, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    BW, NW = xw.shape[:2]
    qkv = F.linear(xw, sd[f"{p}.qkv.weight"]).reshape(BW, NW, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((C // nh) ** -0.5)
    attn = q @ k.transpose(-2, -1)
    table = sd[f"{p}.relative_position_bias_table"]
    index = sd[f"{p}.relative_position_index"]
    rpb = table[index.view(-1)].view(ws * ws, ws * ws, -1).permute(2, 0, 1).contiguous()
    attn = (attn + rpb.unsqueeze(0)).softmax(dim=-1)
    xw = (attn @ v).transpose(1, 2).reshape(BW, NW, C)
    xw = F.linear(xw, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])
    y = xw.view(B_, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B_, H, W, C)
    return y.reshape(B_, H * W, C)


def feed_forward(sd, p, x):
    """reference attention.py:153-168: nn.LayerNorm -> Linear(d, 2*inner, no bias) -> GEGLU
    (value = first half, gate = second half, exact erf GELU) -> Linear(inner, d, no bias)."""
    xn = layer_norm(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"])
    hdn = F.linear(xn, sd[f"{p}.1.weight"])
    val, gate = hdn.chunk(2, dim=-1)
    return F.linear(F.gelu(gate) * val, sd[f"{p}.4.weight"])


def pooling(sd, p, x, kind):
    """reference attention.py:83-113 Pooling.forward on [B, N, C] tokens of a sqrt(N) x sqrt(N) grid:
    'a' AvgPool2d(2), 'm' MaxPool2d(2); 'l' Linear(4C -> C) on x.view(B, N/4, 4C), i.e. on four
    CONSECUTIVE tokens of the row-major sequence (not a 2x2 window)."""
    B, N, C = x.shape
    if kind in "am":
        H = W = int(math.sqrt(N))
        y = x.view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        y = F.avg_pool2d(y, 2) if kind == "a" else F.max_pool2d(y, 2)
        return y.view(B, C, -1).transpose(1, 2).contiguous()
    return F.linear(x.reshape(B, N // 4, 4 * C), sd[f"{p}.pool.weight"], sd[f"{p}.pool.bias"])


def up(sd, p, x, kind):
    """reference attention.py:116-150 Up.forward on [B, N, C] tokens of a sqrt(N) x sqrt(N) grid:
    'n' nn.Upsample(2, 'nearest') -> [B, 4N, C]; 'r' the same followed by Linear(C, C) (+bias)."""
    B, N, C = x.shape
    H = W = int(math.sqrt(N))
    y = x.view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    y = F.interpolate(y, scale_factor=2, mode="nearest")
    y = y.view(B, C, -1).transpose(1, 2).contiguous()  # 'n': :142; 'r': Rearrange('b c h w -> b (h w) c')
    if kind == "r":
        y = F.linear(y, sd[f"{p}.up.2.weight"], sd[f"{p}.up.2.bias"])
    return y


def transformer(sd, prefix, x, block, video_shape, cfg, is_spatial, causal, spatial_pos, taps=None):
    """reference attention.py:655-689 Transformer.forward: per block PEG(+res) -> attention(+res)
    -> FF(+res); final custom LayerNorm."""
    for i, c in enumerate(block):
        p = f"{prefix}.layers.{i}"
        if c == "t":
            x = peg(x, sd[f"{p}.0.dsconv.weight"], sd[f"{p}.0.dsconv.bias"], video_shape,
                    cfg.causal_in_peg) + x
            x = attention(sd, f"{p}.1", x, cfg, is_spatial, causal, spatial_pos) + x
        elif c == "w":
            x = window_attention(sd, f"{p}.1", x, cfg) + x
        elif c in "aml":
            x = pooling(sd, f"{p}.1", x, c)  # no residual, attention.py:674
        elif c in "nr":
            # Up blocks, no residual (attention.py:674). Encoder only: the reference DECODER raises on them
            # (einops shape mismatch at omnitokenizer.py:1078, probed), decode() below rejects them
            x = up(sd, f"{p}.1", x, c)
        else:
            raise NotImplementedError(c)
        x = feed_forward(sd, f"{p}.3", x) + x
        if c in "aml":  # attention.py:683-684
            video_shape = (video_shape[0], video_shape[1], video_shape[2] // 2, video_shape[3] // 2)
        elif c in "nr":  # attention.py:686-687
            video_shape = (video_shape[0], video_shape[1], video_shape[2] * 2, video_shape[3] * 2)
        if taps is not None:
            taps[f"{prefix}.layers.{i}"] = x
    return layer_norm(x, sd[f"{prefix}.norm_out.gamma"], sd[f"{prefix}.norm_out.beta"])


# --------------------------------------------------------------------------------------------
# encoder / decoder / VQ
# --------------------------------------------------------------------------------------------


def patchify(frames, p, pt):
    """einops 'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)' (reference omnitokenizer.py:807,
    815); pt=1 for the first frame."""
    B, C, Fr, H, W = frames.shape
    t, h, w = Fr // pt, H // p, W // p
    x = frames.reshape(B, C, t, pt, h, p, w, p).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(B, t, h, w, C * pt * p * p)


def unpatchify(tok, C, p, pt):
    """einops 'b t h w (c pt p1 p2) -> b c (t pt) (h p1) (w p2)' (omnitokenizer.py:1008,1015)."""
    B, t, h, w, _ = tok.shape
    x = tok.reshape(B, t, h, w, C, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7)
    return x.reshape(B, C, t * pt, h * p, w * p)


def patch_embed(sd, video, cfg):
    """reference omnitokenizer.py:806-822, 934-945: separate weights for frame 0 and the rest."""
    p_enc, pt_enc = cfg.enc_patch_size, cfg.enc_temporal_patch_size

    def emb(name, frames, pt):
        if cfg.patch_embed == "cnn":
            # :823-838 Conv3d(kernel = stride = (pt, p, p)) -> SyncBatchNorm (eval: running stats)
            n = f"encoder.{name}"
            t = F.conv3d(frames, sd[f"{n}.0.weight"], sd[f"{n}.0.bias"], stride=(pt, p_enc, p_enc))
            t = F.batch_norm(t, sd[f"{n}.1.running_mean"], sd[f"{n}.1.running_var"], sd[f"{n}.1.weight"],
                             sd[f"{n}.1.bias"], False, 0.1, 1e-5)
            return t.permute(0, 2, 3, 4, 1)
        t = patchify(frames, p_enc, pt)
        t = layer_norm(t, sd[f"encoder.{name}.1.weight"], sd[f"encoder.{name}.1.bias"])
        t = F.linear(t, sd[f"encoder.{name}.2.weight"], sd[f"encoder.{name}.2.bias"])
        return layer_norm(t, sd[f"encoder.{name}.3.weight"], sd[f"encoder.{name}.3.bias"])
    f = video.shape[2]
    assert (f - 1) % pt_enc == 0, \
        f"number of frames ({f}) minus one must be divisible by temporal patch size"
    tok = emb("to_patch_emb_first_frame", video[:, :, :1], 1)
    if f > 1:
        tok = torch.cat([tok, emb("to_patch_emb", video[:, :, 1:], pt_enc)], dim=1)
    return tok  # b t h w d


def encoder(sd, x, is_image, cfg, taps=None):
    """reference omnitokenizer.py:919-947 + 881-916: returns tokens [b, t, h, w, d] (channel-last;
    the reference returns 'b d t h w' and pre_vq_conv immediately rearranges it back)."""
    video = x.unsqueeze(2) if is_image else x
    tok = patch_embed(sd, video, cfg)
    if taps is not None:
        taps["patch_embed"] = tok
    b, t, h, w, d = tok.shape
    shape = (b, t, h, w)
    s = tok.reshape(b * t, h * w, d)
    s = transformer(sd, "encoder.enc_spatial_transformer", s, cfg.enc_block, shape, cfg,
                    True, False, cfg.spatial_pos, taps)
    if taps is not None:
        taps["enc_spatial"] = s
    h = w = int(math.sqrt(s.shape[1]))  # :898-899 (pooling blocks shrink the grid)
    shape = (b, t, h, w)
    s = s.reshape(b, t, h, w, d).permute(0, 2, 3, 1, 4).reshape(b * h * w, t, d)
    s = transformer(sd, "encoder.enc_temporal_transformer", s, "t" * cfg.temporal_depth, shape, cfg,
                    False, cfg.causal_in_temporal_transformer, "rel", taps)
    tok = s.reshape(b, h, w, t, d).permute(0, 3, 1, 2, 4).contiguous()
    if cfg.patch_embed == "linear" and (cfg.defer_spatial_pool or cfg.defer_temporal_pool):
        y = tok.permute(0, 4, 1, 2, 3)  # b d t h w, :907-914
        if cfg.defer_spatial_pool:
            y = F.avg_pool3d(y, (1, 2, 2))
        if cfg.defer_temporal_pool and y.shape[2] > 1:
            y = torch.cat([y[:, :, :1], F.avg_pool3d(y[:, :, 1:], (2, 1, 1))], dim=2)
        tok = y.permute(0, 2, 3, 4, 1).contiguous()
    return tok


def pre_vq(sd, tok, cfg):
    """reference omnitokenizer.py:143-148 Linear(dim->codebook_dim)+bias on channel-last, then
    F.normalize over the channel dim (:251-252, l2_code).  Returns z [b,t,h,w,cdim]."""
    z = F.linear(tok, sd["pre_vq_conv.1.weight"], sd["pre_vq_conv.1.bias"])
    if cfg.l2_code:
        z = F.normalize(z, p=2, dim=-1)
    return z


def vq_argmin(z_flat, codebook):
    """reference modules/codebook.py:82-86, same expression: d = sum(x^2) - (2x)@E^T + sum(E^2),
    argmin -> first minimum.  oracle/vq_argmin.c restates the arithmetic explicitly (fmaf chain)."""
    dist = (z_flat ** 2).sum(dim=1, keepdim=True) - 2 * z_flat @ codebook.t() \
        + (codebook.t() ** 2).sum(dim=0, keepdim=True)
    return torch.argmin(dist, dim=1)


def codebook_usage_update(ids, n_codes, usage_prev, call_cnt, sigma=0.99):
    """The statistics AND the eval-time state mutation of reference Codebook.forward (modules/codebook.py:54-72
    calculate_batch_codebook_usage_percentage, :122-123 perplexity, :133-140 EMA buffer / call_cnt / avg_usage).
    Returns (batch_usage [n_codes], perplexity, avg_usage, new codebook_usage buffer, new call_cnt)."""
    flat = ids.reshape(-1)
    total = flat.numel()
    usage = torch.zeros(n_codes, dtype=torch.float32)
    uniq, counts = torch.unique(flat, return_counts=True)            # :64
    usage[uniq.long()] = counts.float() / total                      # :66-69
    avg_probs = torch.bincount(flat, minlength=n_codes).float() / total  # mean of the one-hot rows, :87, :122
    perplexity = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))  # :123
    new = usage if call_cnt == 0 else sigma * usage_prev + (1 - sigma) * usage   # :133-136
    avg_usage = (new > (1 / n_codes)).sum() / n_codes                # :140
    return usage, perplexity, avg_usage, new, call_cnt + 1


def ext_pre_vq(sd, tok, cosine=True):
    """--use_external_codebook: VectorQuantize.project_in (512 -> codebook_dim, +bias) then the codebook's
    transform_input: l2norm for the cosine codebook, identity for the Euclidean one
    (vector_quantize_pytorch.py:905, 915, 535, 259).  [b,t,h,w,cdim]."""
    z = F.linear(tok, sd["codebook.project_in.weight"], sd["codebook.project_in.bias"])
    return F.normalize(z, p=2, dim=-1) if cosine else z


def vq_argmin_cdist(z_flat, embed):
    """EuclideanCodebook.forward in eval (vector_quantize_pytorch.py:29-33, 463-465): dist = -cdist,
    cdist = sqrt(clamp(x2 + y2 + (-2) * x.y, 0)); ids = argmax -> first maximum."""
    x2 = (z_flat ** 2).sum(-1)
    y2 = (embed ** 2).sum(-1)
    xy = torch.einsum("bid,bjd->bij", z_flat[None], embed[None])[0] * -2
    dist = -(x2[:, None] + y2[None, :] + xy).clamp(min=0).sqrt()
    return dist.argmax(dim=-1)


def vq_argmax_cos(z_flat, embed):
    """CosineSimCodebook.forward in eval (vector_quantize_pytorch.py:646-650): dist = einsum('h n d, h c d
    -> h n c'), ids = argmax (gumbel_sample at temperature 0 / eval) -> first maximum."""
    dist = torch.einsum("hnd,hcd->hnc", z_flat[None], embed[None])[0]
    return dist.argmax(dim=-1)


def ext_embeddings(sd, ids):
    """quantize = embed[ids] -> project_out (vector_quantize_pytorch.py:653, 1070): tokens [..., 512]."""
    q = F.embedding(ids, sd["codebook._codebook.embed"][0])
    return F.linear(q, sd["codebook.project_out.weight"], sd["codebook.project_out.bias"])


def encode(sd, x, is_image, cfg, include_embeddings=False, taps=None):
    """reference omnitokenizer.py:247-258 VQGAN.encode."""
    tok = encoder(sd, x, is_image, cfg, taps)
    if cfg.use_external_codebook:
        z = ext_pre_vq(sd, tok, cfg.l2_code)
        if taps is not None:
            taps["z"] = z
        b, t, h, w, c = z.shape
        quant = vq_argmax_cos if cfg.l2_code else vq_argmin_cdist  # use_cosine_sim = args.l2_code, :134
        ids = quant(z.reshape(-1, c), sd["codebook._codebook.embed"][0]).view(b, t, h, w)
        if include_embeddings:  # eval: no straight-through term (vector_quantize_pytorch.py:935)
            return ext_embeddings(sd, ids).permute(0, 4, 1, 2, 3), ids
        return ids
    z = pre_vq(sd, tok, cfg)
    if taps is not None:
        taps["z"] = z
    b, t, h, w, c = z.shape
    ids = vq_argmin(z.reshape(-1, c), sd["codebook.embeddings"]).view(b, t, h, w)
    if include_embeddings:
        emb = F.embedding(ids, sd["codebook.embeddings"]).permute(0, 4, 1, 2, 3)
        zc = z.permute(0, 4, 1, 2, 3)
        return (emb - zc) + zc, ids  # codebook.py:120 straight-through value
    return ids


def vae_posterior(sd, tok):
    """reference omnitokenizer.py:248 + modules/vae.py:4-13: pre_vq Linear(dim -> 2*cdim) on channel-last
    tokens, mean | logvar = chunk over channels, logvar clamped to [-30, 20], std = exp(0.5 logvar).
    Returns (mean, std) as [b, c, t, h, w] like the reference's 'b c t h w' tensors."""
    h = F.linear(tok, sd["pre_vq_conv.1.weight"], sd["pre_vq_conv.1.bias"]).permute(0, 4, 1, 2, 3)
    mean, logvar = torch.chunk(h, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean, torch.exp(0.5 * logvar)


def encode_vae(sd, x, is_image, cfg, noise=None):
    """reference omnitokenizer.py:260-266 (--use_vae): z = mean + std * randn (vae.py:15-17; the
    reference draws the noise on the CPU with the global generator).  noise [b,c,t,h,w] or None."""
    tok = encoder(sd, x, is_image, cfg)
    mean, std = vae_posterior(sd, tok)
    if noise is None:
        noise = torch.randn(mean.shape)
    z = mean + std * noise
    return z.squeeze(2) if is_image else z


def decode_vae(sd, z, is_image, cfg):
    """reference omnitokenizer.py:293-317 (--use_vae) -> post_vq -> decoder.  Accepted latent
    layouts are the reference's, which are NOT symmetric with encode's output for video:
      image: [b, hw, c] (:298-301) or channel-first [b, c, h, w] (:303-304, what encode returns);
      video: [b, t*h*w, c] (:309-312) or channel-LAST [b, t, h, w, c] (:313-314; Latte
             sample_ddp.py:201-203 permutes the sampler's output into it) -- encode's
             [b, c, t, h, w] must be permuted by the caller."""
    if z.ndim == 3:
        if is_image:
            hh = int(math.sqrt(z.shape[1]))
            zl = z.reshape(z.shape[0], 1, hh, -1, z.shape[-1])
        else:
            hh = cfg.resolution // cfg.patch_size
            zl = z.reshape(z.shape[0], -1, hh, hh, z.shape[-1])
    elif is_image:
        assert z.ndim == 4
        zl = z.permute(0, 2, 3, 1).unsqueeze(1)
    else:
        assert z.ndim == 5
        zl = z
    tok = F.linear(zl, sd["post_vq_conv.1.weight"], sd["post_vq_conv.1.bias"])
    return _decode_tokens(sd, tok, is_image, cfg, None)


def decode(sd, ids, is_image, cfg, taps=None):
    """reference omnitokenizer.py:268-291 VQGAN.decode + 1101-1118 / 1059-1098 decoder.
    --use_external_codebook: the reference's decode() raises (it reads self.codebook.embeddings, which
    VectorQuantize does not have); what its forward() computes on the same ids is restated instead:
    decoder(project_out(embed[ids])) (omnitokenizer.py:362-363 with post_vq_conv = Identity)."""
    if cfg.use_external_codebook:
        if ids.ndim == 2:
            hh = int(math.sqrt(ids.shape[1])) if is_image else cfg.resolution // cfg.patch_size
            ids = ids.reshape(ids.shape[0], -1, hh, hh) if not is_image else ids.reshape(ids.shape[0], 1, hh, -1)
        return _decode_tokens(sd, ext_embeddings(sd, ids), is_image, cfg, taps)
    z = F.embedding(ids, sd["codebook.embeddings"])
    if z.ndim == 3:
        if is_image:
            hh = int(math.sqrt(z.shape[1]))
            z = z.reshape(z.shape[0], 1, hh, -1, z.shape[-1])
        else:
            hh = cfg.resolution // cfg.patch_size
            z = z.reshape(z.shape[0], -1, hh, hh, z.shape[-1])
    tok = F.linear(z, sd["post_vq_conv.1.weight"], sd["post_vq_conv.1.bias"])  # b t h w d
    if taps is not None:
        taps["post_vq"] = tok
    return _decode_tokens(sd, tok, is_image, cfg, taps)


def _decode_tokens(sd, tok, is_image, cfg, taps):
    """decoder on post_vq tokens [b,t,h,w,d] (reference omnitokenizer.py:1101-1118 / 1059-1098)."""
    if cfg.patch_embed == "linear" and (cfg.defer_spatial_pool or cfg.defer_temporal_pool):
        y = tok.permute(0, 4, 1, 2, 3)  # b d t h w, omnitokenizer.py:1101-1110
        if cfg.defer_temporal_pool and y.shape[2] > 1:
            y = torch.cat([y[:, :, :1], F.interpolate(y[:, :, 1:], scale_factor=(2, 1, 1), mode="nearest")], dim=2)
        if cfg.defer_spatial_pool:
            y = F.interpolate(y, scale_factor=(1, 2, 2), mode="nearest")
        tok = y.permute(0, 2, 3, 4, 1).contiguous()
    b, t, h, w, d = tok.shape
    shape = (b, t, h, w)
    s = tok.permute(0, 2, 3, 1, 4).reshape(b * h * w, t, d)
    s = transformer(sd, "decoder.dec_temporal_transformer", s, "t" * cfg.temporal_depth, shape, cfg,
                    False, cfg.causal_in_temporal_transformer, "rel", taps)
    s = s.reshape(b, h, w, t, d).permute(0, 3, 1, 2, 4).reshape(b * t, h * w, d)
    s = transformer(sd, "decoder.dec_spatial_transformer", s, cfg.dec_block, shape, cfg,
                    True, False, cfg.spatial_pos, taps)
    s = s.reshape(b, t, h, w, d)
    if taps is not None:
        taps["dec_tokens"] = s
    C, p, pt = cfg.image_channels, cfg.dec_patch_size, cfg.dec_temporal_patch_size

    def pixels(name, toks, ptk):
        n = f"decoder.{name}"
        if cfg.patch_embed == "cnn":
            # :1019-1033 ConvTranspose3d(kernel = stride) -> SyncBatchNorm(3) (eval)
            y = F.conv_transpose3d(toks.permute(0, 4, 1, 2, 3), sd[f"{n}.1.weight"], sd[f"{n}.1.bias"],
                                   stride=(ptk, p, p))
            return F.batch_norm(y, sd[f"{n}.2.running_mean"], sd[f"{n}.2.running_var"], sd[f"{n}.2.weight"],
                                sd[f"{n}.2.bias"], False, 0.1, 1e-5)
        return unpatchify(F.linear(toks, sd[f"{n}.0.weight"], sd[f"{n}.0.bias"]), C, p, ptk)

    out = pixels("to_pixels_first_frame", s[:, :1], 1)
    if t > 1:
        out = torch.cat([out, pixels("to_pixels", s[:, 1:], pt)], dim=2)
    return out[:, :, 0] if is_image else out


def psnr(a, b):
    """reference evaluation/common_metrics_on_video_quality/calculate_psnr.py:6-15 on
    clamp(x+0.5,0,1): [0,1] range, peak 1, mse floor 1e-10 -> 100 dB."""
    a = torch.clamp(a.double() + 0.5, 0, 1)
    b = torch.clamp(b.double() + 0.5, 0, 1)
    mse = torch.mean((a - b) ** 2).item()
    if mse < 1e-10:
        return 100.0
    return 20 * math.log10(1.0 / math.sqrt(mse))
