"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the LM consumer of the token path.

A functional fp32 restatement (plain torch CPU ops on a flat state_dict) of the reference's GPT
(OmniTokenizer/modules/gpt.py): the full-sequence forward, the KV-cached one-token step and the
sampling loops of `sample_with_past` / `sample_with_past_cfg`.  Every function cites the reference
lines it follows.  Only tests/ and benchmarks' cpu_baseline legs may import this; the product
(omnitokenizer_amd/) never does.

Pinning: tests/golden/gpt_*.npz hold logits / greedy samples of the reference's own GPT class
(imported unmodified by tests/golden/make_golden.py) on seeded weights; tests/test_oracle_gpt.py
checks this file against them and, when /root/reference is present, against the live class.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def attention(sd, p, x, n_head, past_k=None, past_v=None):
    """reference gpt.py:108-140 CausalSelfAttention.forward on x [B,T,C]; with a past the new tokens
    attend to everything (is_causal = layer_past is None, :125), scale 1/sqrt(head_dim)."""
    B, T, C = x.shape
    hs = C // n_head
    k = F.linear(x, sd[f"{p}.key.weight"], sd[f"{p}.key.bias"]).view(B, T, n_head, hs).transpose(1, 2)
    q = F.linear(x, sd[f"{p}.query.weight"], sd[f"{p}.query.bias"]).view(B, T, n_head, hs).transpose(1, 2)
    v = F.linear(x, sd[f"{p}.value.weight"], sd[f"{p}.value.bias"]).view(B, T, n_head, hs).transpose(1, 2)
    new_k, new_v = k, v
    if past_k is not None:
        k = torch.cat((past_k, k), dim=-2)
        v = torch.cat((past_v, v), dim=-2)
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs))
    if past_k is None:
        mask = torch.tril(torch.ones(T, T, dtype=torch.bool))
        att = att.masked_fill(~mask, float("-inf"))
    y = F.softmax(att, dim=-1) @ v
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    return F.linear(y, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"]), new_k, new_v


def block(sd, p, x, n_head, past_k=None, past_v=None):
    """reference gpt.py:143-167 Block: x + attn(ln1(x)); x + mlp(ln2(x)), exact-erf GELU."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[f"{p}.ln1.weight"], sd[f"{p}.ln1.bias"])
    a, k, v = attention(sd, f"{p}.attn", h, n_head, past_k, past_v)
    x = x + a
    h = F.layer_norm(x, (C,), sd[f"{p}.ln2.weight"], sd[f"{p}.ln2.bias"])
    h = F.gelu(F.linear(h, sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"]))
    return x + F.linear(h, sd[f"{p}.mlp.2.weight"], sd[f"{p}.mlp.2.bias"]), k, v


def n_layers(sd):
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))


def vtokens_position_embeddings(sd, cbox, tbox=None):
    """reference gpt.py:220-225: [B, n, C] rows of vtokens_pos_emb [1, T, H, W, C] selected by the per-sample
    spatial boxes cbox = (h0, h1, w0, w1) and optional temporal boxes tbox = (t0, t1)."""
    e = sd["vtokens_pos_emb"]
    C = e.shape[-1]
    if tbox:
        return torch.cat([e[:, tp[0]:tp[1], p[0]:p[1], p[2]:p[3], :].reshape(1, -1, C) for p, tp in zip(cbox, tbox)], 0)
    return torch.cat([e[:, :, p[0]:p[1], p[2]:p[3], :].reshape(1, -1, C) for p in cbox], 0)


def forward(sd, idx, n_head, embeddings=None, cbox=None, tbox=None):
    """reference gpt.py:207-234 GPT.forward(idx, embeddings, cbox=, tbox=) -> logits [B,T,V]: explicit
    embeddings are prepended (:214-216); with vtokens_pos (a 'vtokens_pos_emb' entry in sd) the box embeddings
    are added to the position embeddings (:219-226)."""
    tok = F.embedding(idx, sd["tok_emb.weight"])
    if embeddings is not None:
        tok = torch.cat((embeddings, tok), dim=1)
    T = tok.shape[1]
    pos = sd["pos_emb"][:, :T]
    if "vtokens_pos_emb" in sd:
        pos = pos + vtokens_position_embeddings(sd, cbox, tbox)
    x = tok + pos
    for i in range(n_layers(sd)):
        x, _, _ = block(sd, f"blocks.{i}", x, n_head)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd["ln_f.weight"], sd["ln_f.bias"])
    return F.linear(x, sd["head.weight"])


def forward_with_past(sd, idx, n_head, cache, position=None, embeddings=None, cbox=None, past_length=None):
    """reference gpt.py:236-275.  cache: None (first call, positions 0..T-1) or a list of (k, v)
    per layer [B,nh,len,hs]; then idx is one new token per row and its position embedding is
    pos_emb[:, position] (position = past_length, or past_length + 1 with forward_uncond, :248).
    embeddings (first call only) are prepended (:239-240); with vtokens_pos the box embeddings of
    positions [:T] (first call, :255-257) or of index past_length (:249-252) are added.
    Returns (logits [B,T,V], new cache)."""
    tok = F.embedding(idx, sd["tok_emb.weight"])
    if embeddings is not None:
        assert cache is None
        tok = torch.cat((embeddings, tok), dim=1)
    T = tok.shape[1]
    if cache is None:
        pos = sd["pos_emb"][:, :T]
        if "vtokens_pos_emb" in sd:
            pos = pos + vtokens_position_embeddings(sd, cbox)[:, :T]
    else:
        assert T == 1 and position is not None
        pos = sd["pos_emb"][:, position][:, None]
        if "vtokens_pos_emb" in sd:
            pl = position if past_length is None else past_length
            pos = pos + vtokens_position_embeddings(sd, cbox)[:, pl][:, None]
    x = tok + pos
    new_cache = []
    for i in range(n_layers(sd)):
        pk, pv = (None, None) if cache is None else cache[i]
        x, k, v = block(sd, f"blocks.{i}", x, n_head, pk, pv)
        new_cache.append((k, v) if cache is None else (torch.cat((pk, k), -2), torch.cat((pv, v), -2)))
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd["ln_f.weight"], sd["ln_f.bias"])
    return F.linear(x, sd["head.weight"]), new_cache


def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float("inf")):
    """reference gpt.py:19-51 (min_tokens_to_keep = 1)."""
    logits = logits.clone()
    if top_k > 0:
        top_k = min(max(top_k, 1), logits.size(-1))
        logits[logits < torch.topk(logits, top_k)[0][..., -1, None]] = filter_value
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits[remove.scatter(1, sorted_indices, remove)] = filter_value
    return logits


def _pick(logits, sample_logits, top_k, top_p, generator):
    if top_k is not None:
        logits = top_k_top_p_filtering(logits, top_k=top_k, top_p=1.0 if top_p is None else top_p)
    probs = F.softmax(logits, dim=-1)
    if not sample_logits:
        return torch.topk(probs, k=1, dim=-1)[1]
    return torch.multinomial(probs, num_samples=1, generator=generator)


def select_inverse_cdf(values, top_k, top_p, u):
    """What csrc/lm_select.hip computes for ONE row, restated in float64 on top of the reference's own filter:
    the survivors of top_k_top_p_filtering(values) in descending value order (ties by ascending index) -- or, for
    top_k None (no filtering), the whole vocabulary in INDEX order --, their softmax, and the token at which the
    cumulative distribution first exceeds u.
    Returns (token, order, cdf): order = survivor indices by rank, cdf = cumulative probabilities (float64)."""
    v = values.clone().float()
    if top_k is not None:
        v = top_k_top_p_filtering(v[None], top_k=top_k, top_p=1.0 if top_p is None else top_p)[0]
    keep = torch.nonzero(torch.isfinite(v))[:, 0]
    order = keep if top_k is None else keep[torch.sort(-v[keep].double(), stable=True)[1]]
    pr = torch.softmax(v[order].double(), 0)
    cdf = torch.cumsum(pr, 0)
    r = int(torch.searchsorted(cdf, torch.tensor(float(u), dtype=torch.float64), right=True))
    return int(order[min(r, order.numel() - 1)]), order, cdf


def sample_with_past(sd, x, n_head, steps, temperature=1.0, sample_logits=True, top_k=None, top_p=None,
                     generator=None, return_logits=False, cbox=None):
    """reference gpt.py:327-359: x [B, cond_len] conditioning -> [B, steps] new tokens."""
    cond_len = x.shape[1]
    cache, out, all_logits = None, [], []
    for n in range(steps):
        logits, cache = forward_with_past(sd, x, n_head, cache, position=n + cond_len - 1, cbox=cbox)
        logits = logits[:, -1, :] / temperature
        all_logits.append(logits)
        x = _pick(logits, sample_logits, top_k, top_p, generator)
        out.append(x)
    out = torch.cat(out, dim=1)
    return (out, torch.stack(all_logits, 1)) if return_logits else out


def sample_with_past_cfg(sd, x, n_head, steps, temperature=1.0, sample_logits=True, top_k=None, top_p=None,
                         cfg_ratio=1.5, class_first=False, scale_cfg=False, generator=None, return_logits=False):
    """reference gpt.py:387-444 classifier-free guidance: conditional stream [class+1, sos] (or
    [sos, class+1]) and an unconditional stream [sos]; the unconditional stream's new tokens use
    position past_length + 1 (forward_uncond, gpt.py:248) so both streams share positions."""
    x = x + 1
    sos = torch.zeros_like(x)
    xc = torch.cat((x, sos), 1) if class_first else torch.cat((sos, x), 1)
    cond_len = xc.shape[1]
    xu = sos
    cache_c = cache_u = None
    out, all_logits = [], []
    for n in range(steps):
        ratio = n if scale_cfg else 1
        lc, cache_c = forward_with_past(sd, xc, n_head, cache_c, position=n + cond_len - 1)
        lu, cache_u = forward_with_past(sd, xu, n_head, cache_u, position=n + cond_len - 2 + 1)
        lc, lu = lc[:, -1, :] / temperature, lu[:, -1, :] / temperature
        t = cfg_ratio * ratio
        blend = (1 + t) * lc - t * lu
        all_logits.append(blend)
        xc = xu = _pick(blend, sample_logits, top_k, top_p, generator)
        out.append(xc)
    out = torch.cat(out, dim=1)
    return (out, torch.stack(all_logits, 1)) if return_logits else out


# seeded synthetic GPT weights live with the other generators in the product package (pure numpy; the
# golden fixtures pin their crc); re-exported here for the tests
from omnitokenizer_amd.synth import synth_gpt_state  # noqa: E402,F401
