"""Drop-in for the reference's `GPT` (OmniTokenizer/modules/gpt.py:170-275) on the KV-cached
sampling path, plus its sampling loops `sample_with_past` (:327-359) and `sample_with_past_cfg`
(:387-444) -- the LM consumer downstream of `OmniTokenizer_VQGAN.encode()` and upstream of
`decode()` (lm_transformer.py:262,434).

The class owns the parameters under the reference's state_dict key names; all arithmetic of a decode
step runs in libomnitok.so (include/omnitok_lm.h, csrc/lm.hip): a preallocated K/V cache instead of
the reference's per-step torch.cat of all pasts, GEMV kernels that stream each fp32 weight matrix
once per step, flash-decode attention, and -- because the step's launch sequence does not depend on
the position -- one captured HIP graph replayed per token.  Token selection (temperature, the CFG blend,
top-k / top-p filtering, argmax or one multinomial draw) is one more kernel (csrc/lm_select.hip); the only
thing torch contributes is the uniform random number per stream (so torch.manual_seed governs the samples).
Inference only; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import OmnitokLmConfig, check


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; call GPT.forward / forward_with_past")


class _PastHandle:
    """What forward_with_past returns in place of the reference's stacked K/V tensor: the K/V live
    in the engine's cache; the handle only identifies the stream so that the reference's calling
    pattern (`past.append(present)`, gpt.py:343-346) keeps working."""

    def __init__(self, model, length, generation):
        self.model, self.length, self.generation = model, length, generation


class GPT(nn.Module):
    def __init__(self, args, vocab_size, block_size, n_layer=12, n_head=8, n_embd=256, embd_pdrop=0.,
                 resid_pdrop=0., attn_pdrop=0., n_unmasked=0, vtokens_pos=False):
        """Same signature as the reference (gpt.py:172).  Dropouts are inference no-ops; vtokens_pos adds the
        reference's vtokens_pos_emb parameter [1, sequence_length, resolution, resolution, n_embd] (gpt.py:183-184).
        n_unmasked is accepted and has no effect, exactly as in the reference under torch >= 2.1: it only edits the
        `mask` buffer (gpt.py:98-100), which the scaled_dot_product_attention branch the reference then takes never
        reads (gpt.py:122-126, is_causal = layer_past is None)."""
        super().__init__()
        self.n_unmasked = n_unmasked
        self.vtokens_pos = bool(vtokens_pos)
        if self.vtokens_pos:
            self.vtokens_pos_emb = nn.Parameter(
                torch.zeros(1, args.sequence_length, args.resolution, args.resolution, n_embd), requires_grad=False)
        self.block_size = block_size
        self.vocab_size, self.n_layer, self.n_head, self.n_embd = vocab_size, n_layer, n_head, n_embd

        class _Cfg:
            pass
        self.config = _Cfg()
        self.config.vocab_size, self.config.block_size = vocab_size, block_size
        self.config.n_layer, self.config.n_head, self.config.n_embd = n_layer, n_head, n_embd
        C = n_embd
        self.pos_emb = nn.Parameter(torch.zeros(1, block_size, C), requires_grad=False)

        def lin(o, i, bias=True):
            m = _Holder()
            m.weight = nn.Parameter(torch.zeros(o, i), requires_grad=False)
            if bias:
                m.bias = nn.Parameter(torch.zeros(o), requires_grad=False)
            return m

        def ln():
            m = _Holder()
            m.weight = nn.Parameter(torch.ones(C), requires_grad=False)
            m.bias = nn.Parameter(torch.zeros(C), requires_grad=False)
            return m
        self.tok_emb = _Holder()
        self.tok_emb.weight = nn.Parameter(torch.zeros(vocab_size, C), requires_grad=False)
        blocks = []
        for _ in range(n_layer):
            b = _Holder()
            b.ln1, b.ln2 = ln(), ln()
            b.attn = _Holder()
            b.attn.key, b.attn.query, b.attn.value, b.attn.proj = lin(C, C), lin(C, C), lin(C, C), lin(C, C)
            b.mlp = nn.ModuleList([lin(4 * C, C), _Holder(), lin(C, 4 * C)])  # keys mlp.0.* and mlp.2.*
            blocks.append(b)
        self.blocks = nn.ModuleList(blocks)
        self.ln_f = ln()
        self.head = lin(vocab_size, C, bias=False)
        self._engine = None
        self._engine_sig = None
        self._cache_shape = (0, 0)
        self._pos = self._len = None
        self._graphs = {}

    # ---- plumbing ---------------------------------------------------------------------------------
    @property
    def device(self):
        return self.pos_emb.device

    def get_block_size(self):
        return self.block_size

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("omnitokenizer_amd GPT is inference-only")
        return super().train(False)

    def load_state_dict(self, state_dict, strict: Optional[bool] = None, assign: bool = False):
        """Accepts the reference GPT's state_dict; its causal-mask buffers (blocks.N.attn.mask) are
        not parameters of the path and are dropped.  Missing tensors raise unless strict=False is passed
        explicitly (every parameter is pre-created here: a renamed key would otherwise sample from zeros)."""
        sd = {k: v for k, v in state_dict.items() if not k.endswith(".attn.mask")}
        out = super().load_state_dict(sd, strict=bool(strict), assign=assign)
        self._engine_sig = None
        if out.missing_keys:
            msg = f"state_dict lacks {len(out.missing_keys)} tensors of the LM, e.g. {out.missing_keys[:4]}"
            if strict is None:
                raise RuntimeError(msg + " (pass strict=False to load anyway)")
            import warnings
            warnings.warn(msg)
        return out

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.load().omnitok_lm_destroy(self._engine)
        except Exception:
            pass

    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())

    def _sync_engine(self):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(f"GPT is on {dev}: move it to the GPU (.to('cuda')). The MI355X HIP path is the "
                               "only implementation; there is no CPU fallback.")
        sig = self._signature()
        if self._engine is not None and sig == self._engine_sig:
            return
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        if self._engine is None:
            cfg = OmnitokLmConfig(self.vocab_size, self.block_size, self.n_layer, self.n_head, self.n_embd)
            h = ctypes.c_void_p()
            check(lib.omnitok_lm_create(ctypes.byref(cfg), ctypes.byref(h)), "lm_create")
            self._engine = h
        for name, t in self.state_dict(keep_vars=True).items():
            if name == "vtokens_pos_emb":  # gathered per call on the host side (cbox / tbox), not an engine weight
                continue
            t = t.detach()
            if t.dtype != torch.float32:
                raise TypeError(f"{name}: parameters must be float32 (the path computes in fp32 like the reference)")
            t = t.contiguous()
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            check(lib.omnitok_lm_set_weight(self._engine, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim(),
                                            stream), f"lm_set_weight({name})")
        check(lib.omnitok_lm_finalize(self._engine, stream), "lm_finalize")
        torch.cuda.current_stream().synchronize()
        self._engine_sig = sig
        self._graphs = {}

    def _ensure_cache(self, batch, length):
        mb, ml = self._cache_shape
        if batch <= mb and length <= ml:
            return
        mb, ml = max(mb, batch), max(ml, min(max(length, 64), self.block_size + 1))
        check(_lib.load().omnitok_lm_alloc_cache(self._engine, mb, ml), "lm_alloc_cache")
        self._cache_shape = (mb, ml)
        self._pos = torch.zeros(mb, dtype=torch.int32, device=self.device)
        self._len = torch.zeros(mb, dtype=torch.int32, device=self.device)
        self._graphs = {}

    def cache_bytes(self) -> int:
        return 0 if self._engine is None else int(_lib.load().omnitok_lm_cache_bytes(self._engine))

    # ---- native stepping ----------------------------------------------------------------------------
    def reset_streams(self, batch, max_len):
        """Starts `batch` empty streams (cache length 0, position 0) with room for max_len tokens."""
        self._sync_engine()
        if batch > 16:
            raise ValueError("at most 16 streams per engine")
        self._ensure_cache(batch, max_len)
        self._pos.zero_()
        self._len.zero_()
        # handles returned by forward_with_past before this point belong to streams that no longer exist
        self._generation = getattr(self, "_generation", 0) + 1

    def check_overflow(self):
        """Raises if a decode step since the last check ran past the allocated K/V cache (its logits are
        invalid); one host synchronisation -- the sampling loops call it once at their end."""
        if not self._engine:
            return
        rc = _lib.load().omnitok_lm_overflowed(self._engine, torch.cuda.current_stream().cuda_stream)
        if rc < 0:
            check(rc, "lm_overflowed")  # a failed read-back is an error, not "no overflow"
        if rc > 0:
            raise RuntimeError("a stream stepped past the K/V cache length it was allocated with")

    @staticmethod
    def _fp(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    def _f32(self, t, shape, name):
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def step(self, idx, logits=None, advance=True, want_logits=True, emb=None, pos_extra=None):
        """One decode step: idx [B] int64 (device) enters every stream at pos[b] / cache_len[b].
        emb [B, C]: an explicit input vector instead of tok_emb[idx] (idx may be None); pos_extra [B, C]: added to
        the position embedding (the vtokens_pos term).  Returns logits [B, vocab] (or None if want_logits is False)."""
        B = (idx if idx is not None else emb).shape[0]
        if want_logits and logits is None:
            logits = torch.empty(B, self.vocab_size, device=self.device, dtype=torch.float32)
        check(_lib.load().omnitok_lm_step_ex(self._engine, self._fp(idx), self._fp(emb), self._fp(pos_extra),
                                             ctypes.c_void_p(self._pos.data_ptr()), ctypes.c_void_p(self._len.data_ptr()),
                                             B, None if not want_logits else ctypes.c_void_p(logits.data_ptr()),
                                             int(advance), torch.cuda.current_stream().cuda_stream), "lm_step")
        return logits if want_logits else None

    def prefill(self, idx, want_logits=False, emb=None, pos_extra=None):
        """Feeds emb [B, Te, C] (optional, prepended) then idx [B, T] into EMPTY streams in one batched pass
        (GEMMs + causal flash attention, include/omnitok_lm.h omnitok_lm_prefill_ex): same arithmetic as Te + T
        steps.  pos_extra [B, Te + T, C] (optional): the vtokens_pos term.  Returns the teacher-forced logits
        [B, Te + T, vocab] if want_logits."""
        B, T = idx.shape
        Te = 0 if emb is None else emb.shape[1]
        idx = idx.contiguous()
        logits = torch.empty(B, Te + T, self.vocab_size, device=idx.device, dtype=torch.float32) if want_logits else None
        check(_lib.load().omnitok_lm_prefill_ex(self._engine, self._fp(idx), T, self._fp(emb), Te, self._fp(pos_extra),
                                                ctypes.c_void_p(self._pos.data_ptr()),
                                                ctypes.c_void_p(self._len.data_ptr()), B, self._fp(logits),
                                                torch.cuda.current_stream().cuda_stream), "lm_prefill")
        return logits

    def _feed(self, idx, pos_extra=None):
        """Conditioning tokens idx [B, T] into empty streams: batched prefill, or plain steps when
        the prefix is short (or too large for one prefill launch).  pos_extra [B, T, C] optional."""
        B, T = idx.shape
        if T == 0:
            return
        if T >= 8 and B * T <= 65535:
            self.prefill(idx, pos_extra=None if pos_extra is None else pos_extra.contiguous())
        else:
            for t in range(T):
                self.step(idx[:, t].contiguous(), want_logits=False,
                          pos_extra=None if pos_extra is None else pos_extra[:, t].contiguous())

    def vtokens_position_embeddings(self, cbox, tbox=None):
        """reference gpt.py:220-225: rows of vtokens_pos_emb selected by the per-sample boxes, [B, n, C]
        (a gather on the parameter; the engine adds it to the position embeddings)."""
        if not self.vtokens_pos:
            return None
        if cbox is None:
            raise ValueError("this GPT was built with vtokens_pos: cbox is required (reference gpt.py:221-225)")
        C = self.vtokens_pos_emb.shape[-1]
        if tbox:
            rows = [self.vtokens_pos_emb[:, tp[0]:tp[1], p[0]:p[1], p[2]:p[3], :].reshape(1, -1, C)
                    for p, tp in zip(cbox, tbox)]
        else:
            rows = [self.vtokens_pos_emb[:, :, p[0]:p[1], p[2]:p[3], :].reshape(1, -1, C) for p in cbox]
        return torch.cat(rows, 0).contiguous()

    def graph_step_extra(self, B):
        """graph_step with a position-extra input buffer (vtokens_pos): (idx, extra, logits, replay)."""
        key = ("extra", B)
        if key not in self._graphs:
            idx = torch.zeros(B, dtype=torch.int64, device=self.device)
            extra = torch.zeros(B, self.n_embd, device=self.device, dtype=torch.float32)
            logits = torch.empty(B, self.vocab_size, device=self.device, dtype=torch.float32)
            pos0, len0 = self._pos.clone(), self._len.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.step(idx, logits, pos_extra=extra)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step(idx, logits, pos_extra=extra)
            self._pos.copy_(pos0)
            self._len.copy_(len0)
            self._graphs[key] = (idx, extra, logits, g)
        idx, extra, logits, g = self._graphs[key]
        return idx, extra, logits, g.replay

    def graph_step(self, B):
        """(idx_buffer, logits_buffer, replay) for a captured decode step of B streams: write the next
        tokens into idx_buffer, call replay(), read logits_buffer.  The step advances pos / cache_len on
        the device, so replaying it walks down the sequence."""
        if B not in self._graphs:
            idx = torch.zeros(B, dtype=torch.int64, device=self.device)
            logits = torch.empty(B, self.vocab_size, device=self.device, dtype=torch.float32)
            pos0, len0 = self._pos.clone(), self._len.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.step(idx, logits)  # warm-up outside the capture (lazy module loading)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step(idx, logits)
            self._pos.copy_(pos0)  # the warm-up and the capture pass do not count
            self._len.copy_(len0)
            self._graphs[B] = (idx, logits, g)
        idx, logits, g = self._graphs[B]
        return idx, logits, g.replay

    # ---- the reference's interface -----------------------------------------------------------------
    @torch.no_grad()
    def forward(self, idx, embeddings=None, targets=None, cbox=None, tbox=None):
        """reference gpt.py:207-234: logits [B, T, V] of a whole sequence (teacher-forced), computed by
        the batched prefill (same arithmetic as walking the KV-cached step over the T positions)."""
        B, Tt = idx.shape
        Te = 0 if embeddings is None else embeddings.shape[1]
        T = Te + Tt  # explicit embeddings are prepended (gpt.py:214-216)
        assert T <= self.block_size, "Cannot forward, model block size is exhausted."
        self._sync_engine()
        emb = None if embeddings is None else self._f32(embeddings, (B, Te, self.n_embd), "embeddings")
        extra = self.vtokens_position_embeddings(cbox, tbox)
        if extra is not None:
            extra = self._f32(extra, (B, T, self.n_embd), "vtokens position embeddings (cbox / tbox)")
        self.reset_streams(B, T)
        idx = idx.to(self.device).long()
        if B * T <= 65535:
            out = self.prefill(idx, want_logits=True, emb=emb, pos_extra=extra)
        else:
            out = torch.empty(B, T, self.vocab_size, device=self.device, dtype=torch.float32)
            for t in range(T):
                out[:, t] = self.step(None if t < Te else idx[:, t - Te].contiguous(),
                                      emb=None if t >= Te else emb[:, t].contiguous(),
                                      pos_extra=None if extra is None else extra[:, t].contiguous())
        loss = None
        if targets is not None:
            loss = F.cross_entropy(out.view(-1, out.size(-1)), targets.view(-1))
        return out, loss

    @torch.no_grad()
    def forward_with_past(self, idx, embeddings=None, targets=None, past=None, past_length=None, cbox=None,
                          forward_uncond=False):
        """reference gpt.py:236-275.  past=None: idx [B, T] starts new streams (positions 0..T-1).
        Otherwise idx [B, 1] continues them; `past` is the list of handles returned so far and its
        total length must equal past_length like the reference asserts (:246-247).  The new token's
        position embedding is pos_emb[past_length] (+1 with forward_uncond, :248)."""
        idx = idx.to(self.device).long()
        B, T = idx.shape
        self._sync_engine()
        extra = self.vtokens_position_embeddings(cbox)  # [B, n, C] or None
        if past is None:
            self.reset_streams(B, self.block_size + 1)
            if embeddings is not None:
                # explicit embeddings prepended to the prefix (gpt.py:239-240): one batched pass, all logits
                Te = embeddings.shape[1]
                emb = self._f32(embeddings, (B, Te, self.n_embd), "embeddings")
                ex = None if extra is None else extra[:, :Te + T].contiguous()
                out = self.prefill(idx, want_logits=True, emb=emb, pos_extra=ex)
                return out, None, _PastHandle(self, Te + T, self._generation)
            self._feed(idx[:, :T - 1], None if extra is None else extra[:, :T - 1])
            logits = self.step(idx[:, T - 1].contiguous(),
                               pos_extra=None if extra is None else extra[:, T - 1].contiguous())
            out = torch.zeros(B, T, self.vocab_size, device=self.device) if T > 1 else None
            if out is not None:
                out[:, -1] = logits  # callers read logits[:, -1, :] (gpt.py:347)
            return (out if out is not None else logits[:, None]), None, _PastHandle(self, T, self._generation)
        if embeddings is not None:
            raise NotImplementedError("embeddings together with a past: the reference broadcasts ONE position "
                                      "embedding over the several new rows (gpt.py:248); no caller does this")
        assert past_length is not None
        for h in past:
            if not isinstance(h, _PastHandle) or h.model is not self or h.generation != self._generation:
                raise RuntimeError(
                    "forward_with_past: `past` belongs to streams that were replaced by a later past=None call "
                    "(the engine keeps ONE set of K/V streams; two interleaved sequences -- e.g. the reference's "
                    "sample_with_past_cfg calling the conditional and unconditional pass in turn -- must run as "
                    "rows of one batch: use omnitokenizer_amd.gpt.sample_with_past_cfg)")
        have = sum(p.length for p in past)
        assert have == past_length, f"{have} =/= {past_length}"
        assert T == 1
        self._pos[:B] = past_length + (1 if forward_uncond else 0)
        logits = self.step(idx[:, 0].contiguous(),
                           pos_extra=None if extra is None else extra[:, past_length].contiguous())
        return logits[:, None], None, _PastHandle(self, 1, self._generation)


def select_tokens(logits, sample_logits=True, top_k=None, top_p=None, temperature=1.0, logits_uncond=None,
                  cfg_t=0.0, generator=None, return_logits=False, err=None):
    """Token selection of the reference's loops (gpt.py:347-357, CFG blend :428-431) as ONE kernel
    (include/omnitok_lm.h omnitok_lm_select): logits [B, V] (raw, this call divides by temperature) ->
    ids [B] int64.  top_k None: no filtering (like the reference, top_p is then ignored).  Stochastic draws are
    inverse-CDF lookups with one torch uniform per stream.
    A nucleus cut (top_p < 1) over more survivors than the kernel's sort buffer (16384: top_k = 0 or > 16384 on a large
    vocabulary, or a tie at the k-th value that wide) cannot be evaluated: the kernel raises a device flag instead of
    silently decoding greedily.  `err` (int32[1] on the device, zeroed by the caller) accumulates it so that a sampling
    loop reads it back ONCE at its end (check_select_overflow); without `err` this call checks it itself (one host
    synchronisation, only when top_p < 1).  Without a nucleus cut any number of survivors is sampled exactly."""
    if logits.device.type != "cuda" or logits.dtype != torch.float32:
        raise RuntimeError("select_tokens: logits must be float32 on the GPU (no CPU path)")
    logits = logits.contiguous()
    B, V = logits.shape
    out = torch.empty(B, dtype=torch.int64, device=logits.device)
    u = torch.rand(B, device=logits.device, generator=generator) if sample_logits else None
    blend = torch.empty(B, V, device=logits.device, dtype=torch.float32) if return_logits else None
    own_err = err is None
    if own_err:
        err = torch.zeros(1, dtype=torch.int32, device=logits.device)
    lu = None if logits_uncond is None else logits_uncond.contiguous()
    fp = GPT._fp
    # (1 + t) and t enter as fp32 scalars like `(1 + t) * lc - t * lu` on fp32 tensors
    check(_lib.load().omnitok_lm_select(fp(logits), fp(lu), B, V, float(temperature), float(1 + cfg_t), float(cfg_t),
                                        -1 if top_k is None else int(top_k), 1.0 if top_p is None else float(top_p),
                                        int(bool(sample_logits)), fp(u), fp(out), fp(blend), fp(err),
                                        torch.cuda.current_stream().cuda_stream), "lm_select")
    if own_err and sample_logits and top_k is not None and top_p is not None and top_p < 1.0:
        check_select_overflow(err)
    return (out, blend) if return_logits else out


def check_select_overflow(err):
    """Raises if a token selection since `err` was zeroed met a nucleus cut it could not evaluate (select_tokens)."""
    if int(err.item()):
        raise NotImplementedError("nucleus sampling (top_p < 1) over more than 16384 surviving logits: lower top_k")


@torch.no_grad()
def sample_with_past(x, model: GPT, steps, temperature=1., sample_logits=True, top_k=None, top_p=None,
                     callback=None, cbox=None, use_graph=True, return_logits=False):
    """reference gpt.py:327-359: x [B, cond_len] conditioning tokens -> [B, steps] sampled tokens.
    The conditioning is fed through the same KV-cached step; each sampling step is one replay of
    the captured HIP graph plus the token selection on the GPU, with no host synchronisation."""
    x = x.to(model.device).long()
    B, cond_len = x.shape
    if cond_len + steps - 1 > model.block_size:  # the reference fails at pos_emb[:, past_length] (gpt.py:248)
        raise ValueError(f"{cond_len} conditioning + {steps} sampled tokens exceed block_size {model.block_size}")
    model._sync_engine()
    extra = model.vtokens_position_embeddings(cbox)  # [B, n, C] or None (vtokens_pos models need cbox)
    model.reset_streams(B, cond_len + steps)
    model._feed(x[:, :cond_len - 1], None if extra is None else extra[:, :cond_len - 1])
    if use_graph:
        if extra is None:
            idx_buf, logits_buf, replay = model.graph_step(B)
        else:
            idx_buf, extra_buf, logits_buf, replay = model.graph_step_extra(B)
    nxt = x[:, -1].contiguous()
    out = torch.empty(B, steps, dtype=torch.int64, device=model.device)
    all_logits = [] if return_logits else None
    nucleus = bool(sample_logits) and top_k is not None and top_p is not None and top_p < 1.0
    sel_err = torch.zeros(1, dtype=torch.int32, device=model.device)
    for n in range(steps):
        if callback is not None:
            callback(n)
        ex = None if extra is None else extra[:, cond_len - 1 + n].contiguous()  # position of the token entering
        if use_graph:
            idx_buf.copy_(nxt)
            if ex is not None:
                extra_buf.copy_(ex)
            replay()
            logits = logits_buf
        else:
            logits = model.step(nxt, pos_extra=ex)
        sel = select_tokens(logits, sample_logits, top_k, top_p, temperature, return_logits=return_logits, err=sel_err)
        if return_logits:
            nxt, lg = sel
            all_logits.append(lg)
        else:
            nxt = sel
        out[:, n] = nxt
    if nucleus:
        check_select_overflow(sel_err)  # one read-back for the whole loop
    return (out, torch.stack(all_logits, 1)) if return_logits else out


@torch.no_grad()
def sample_with_past_cfg(x, model: GPT, steps, temperature=1., sample_logits=True, top_k=None, top_p=None,
                         callback=None, cbox=None, cfg_ratio=1.5, class_first=False, scale_cfg=False, use_graph=True,
                         return_logits=False):
    """reference gpt.py:387-444 (classifier-free guidance).  The conditional stream ([class+1, sos] or
    [sos, class+1]) and the unconditional stream ([sos]) run as rows [0, B) and [B, 2B) of ONE batched
    step: both see the same new token and the same position (the reference's forward_uncond shift,
    gpt.py:248), only their cache lengths differ by one."""
    if cbox is not None or model.vtokens_pos:
        raise NotImplementedError("classifier-free guidance with vtokens_pos boxes: the reference's unconditional "
                                  "pass indexes the box embeddings at past_length of a different stream "
                                  "(gpt.py:250-252); no script combines the two")
    x = x.to(model.device).long() + 1
    B = x.shape[0]
    sos = torch.zeros_like(x)
    cond = torch.cat((x, sos), 1) if class_first else torch.cat((sos, x), 1)
    cond_len = cond.shape[1]  # 2
    if cond_len + steps - 1 > model.block_size:
        raise ValueError(f"{cond_len} conditioning + {steps} sampled tokens exceed block_size {model.block_size}")
    model.reset_streams(2 * B, cond_len + steps)
    # conditioning prefix of the conditional rows only (rows [B, 2B) stay empty): advance them by hand
    for t in range(cond_len - 1):
        tok = torch.cat((cond[:, t], sos[:, 0])).contiguous()
        model.step(tok, want_logits=False, advance=False)
        model._pos[:B] += 1
        model._len[:B] += 1
    model._pos[B:2 * B] = 0
    nxt = torch.cat((cond[:, -1], sos[:, 0])).contiguous()
    if use_graph:
        idx_buf, logits_buf, replay = model.graph_step(2 * B)
    out = torch.empty(B, steps, dtype=torch.int64, device=model.device)
    all_logits = [] if return_logits else None
    nucleus = bool(sample_logits) and top_k is not None and top_p is not None and top_p < 1.0
    sel_err = torch.zeros(1, dtype=torch.int32, device=model.device)
    for n in range(steps):
        if callback is not None:
            callback(n)
        if n == 1:
            # after the first step the unconditional rows skip position 1 (forward_uncond, gpt.py:248)
            model._pos[B:2 * B] = model._pos[:B]
        if use_graph:
            idx_buf.copy_(nxt)
            replay()
            logits = logits_buf
        else:
            logits = model.step(nxt)
        t = cfg_ratio * (n if scale_cfg else 1)
        sel = select_tokens(logits[:B], sample_logits, top_k, top_p, temperature, logits_uncond=logits[B:], cfg_t=t,
                            return_logits=return_logits, err=sel_err)  # blend = (1 + t) * lc - t * lu inside the kernel
        if return_logits:
            tok, lg = sel
            all_logits.append(lg)
        else:
            tok = sel
        out[:, n] = tok
        nxt = torch.cat((tok, tok)).contiguous()
    if nucleus:
        check_select_overflow(sel_err)
    return (out, torch.stack(all_logits, 1)) if return_logits else out
