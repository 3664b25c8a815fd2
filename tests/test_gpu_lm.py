"""GPU (-m gpu): the LM consumer (include/omnitok_lm.h, omnitokenizer_amd/gpt.py) against the
committed outputs of the reference's GPT class and the CPU oracle."""
import argparse
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gpt_oracle as go
from tests.test_oracle_gpt import GPT_CASES, load_gpt_case

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.fixture(scope="module")
def lib():
    from omnitokenizer_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8, 11])
@pytest.mark.parametrize("N,K", [(1536, 1536), (4608, 1536), (6144, 1536), (1536, 6144), (8192, 1536), (300, 256), (8193, 768),
                                 (2048, 2048), (1000, 8192), (20000, 1536)])
def test_gemv(lib, B, N, K):
    """Every GEMV form against torch: K in {1536, 2048, 6144, 8192} runs the K-sliced kernel for every group of 8 / 4 / 2 / 1 streams
    (6 or 8 waves, one workgroup per CU, ragged row split for N = 1000, more than one workgroup per CU for N = 20000), everything
    else the 4- or 6-wave row kernel."""
    x, w, bias, res = rnd(B, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(B, N, seed=4)
    g, beta = rnd(K, seed=5, scale=0.1) + 1.0, rnd(K, seed=6, scale=0.1)
    s = torch.cuda.current_stream().cuda_stream
    xd, wd, bd, rd, gd, betad = (t.cuda() for t in (x, w, bias, res, g, beta))
    tol = 2e-5 * math.sqrt(K / 1536)
    # plain + bias
    y = torch.empty(B, N, device="cuda")
    assert lib.omnitok_lm_gemv(_p(xd), _p(wd), _p(bd), None, None, None, _p(y), B, N, K, 0, s) == 0
    assert (y.cpu() - F.linear(x, w, bias)).abs().max().item() < tol
    # LayerNorm prologue + GELU epilogue
    assert lib.omnitok_lm_gemv(_p(xd), _p(wd), _p(bd), None, _p(gd), _p(betad), _p(y), B, N, K, 1, s) == 0
    ref = F.gelu(F.linear(F.layer_norm(x, (K,), g, beta), w, bias))
    assert (y.cpu() - ref).abs().max().item() < tol
    # residual, in place
    y = rd.clone()
    assert lib.omnitok_lm_gemv(_p(xd), _p(wd), None, _p(y), None, None, _p(y), B, N, K, 0, s) == 0
    assert (y.cpu() - (F.linear(x, w) + res)).abs().max().item() < tol


@pytest.mark.parametrize("hd", [64, 96, 128])
def test_attn_decode(lib, hd):
    B, H, max_len = 3, 4, 700
    C = H * hd
    nchunk = (max_len + 255) // 256
    s = torch.cuda.current_stream().cuda_stream
    kc = rnd(B, H, max_len, hd, seed=7).cuda()
    vc = rnd(B, H, max_len, hd, seed=8).cuda()
    scratch = torch.empty(B * H * nchunk * (2 + hd), device="cuda")
    out = torch.empty(B, C, device="cuda")
    for lens in ([0, 1, 5], [255, 256, 257], [511, 640, 699]):
        qkv = rnd(B, 3 * C, seed=9 + lens[0])
        cl = torch.tensor(lens, dtype=torch.int32, device="cuda")
        kc0, vc0 = kc.clone(), vc.clone()
        assert lib.omnitok_lm_attn_decode(_p(qkv.cuda()), _p(kc), _p(vc), _p(cl), B, H, hd, max_len, _p(scratch),
                                          _p(out), s) == 0
        q, kn, vn = (t.reshape(B, H, hd) for t in qkv.split(C, dim=1))
        for b, ln in enumerate(lens):
            k = torch.cat([kc0[b, :, :ln].cpu(), kn[b][:, None]], 1)  # H, ln+1, hd
            v = torch.cat([vc0[b, :, :ln].cpu(), vn[b][:, None]], 1)
            att = torch.softmax((q[b][:, None] @ k.transpose(-1, -2)) / math.sqrt(hd), -1)
            ref = (att @ v).reshape(C)
            assert (out[b].cpu() - ref).abs().max().item() < 2e-5, (hd, ln)
            # the new token's K/V were appended at index ln, nothing else changed
            assert torch.equal(kc[b, :, ln].cpu(), kn[b]) and torch.equal(vc[b, :, ln].cpu(), vn[b])
            assert torch.equal(kc[b, :, :ln], kc0[b, :, :ln]) and torch.equal(kc[b, :, ln + 1:], kc0[b, :, ln + 1:])


@pytest.mark.parametrize("B,N,K", [(1, 4608, 1536), (2, 1536, 6144), (1, 8192, 1536), (8, 6144, 1536), (4, 1536, 6144), (8, 1536, 1536)])
def test_gemv_forms_agree(lib, B, N, K):
    """The K-sliced kernel ("lm_ksliced" 2, the default) against the row kernel it replaces: same LayerNorm + GELU +
    residual outputs to rounding (the dot products are summed in a different order)."""
    from omnitokenizer_amd import _lib
    x, w, bias, res = rnd(B, K, seed=1).cuda(), rnd(N, K, seed=2, scale=0.05).cuda(), rnd(N, seed=3).cuda(), rnd(B, N, seed=4).cuda()
    g, beta = (rnd(K, seed=5, scale=0.1) + 1.0).cuda(), rnd(K, seed=6, scale=0.1).cuda()
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        _lib.set_option("lm_mfma", 0)   # (the default; with 1, groups of >= 4 streams take the MFMA kernel whatever "lm_ksliced" says)
        for form in (2, 0):
            _lib.set_option("lm_ksliced", form)
            y = torch.empty(B, N, device="cuda")
            assert lib.omnitok_lm_gemv(_p(x), _p(w), _p(bias), _p(res), _p(g), _p(beta), _p(y), B, N, K, 1, s) == 0
            out[form] = y
    finally:
        _lib.set_option("lm_ksliced", 2)
    assert not torch.equal(out[0], out[2])  # the option is live
    assert float((out[0] - out[2]).abs().max()) < 1e-5 * math.sqrt(K / 1536)


@pytest.mark.parametrize("B", [4, 7, 8, 13, 16])
@pytest.mark.parametrize("N,K", [(1536, 1536), (4608, 1536), (1536, 6144), (8192, 1536), (1000, 1536), (20000, 1536), (2048, 2048), (1000, 8192),
                                 (4000, 6144)])
def test_gemv_mfma_path_equals_valu_path(lib, B, N, K):
    """r06: with "lm_mfma" 1 groups of 4 .. 8 streams take the fp32-MFMA kernel (lm_gemm4_kernel) -- one pass over the weights for all
    of them; the VALU kernels ("lm_mfma" 0, the default: faster, profiles/r06_lm_mfma.txt) serve the same call in groups of 8 / 4 / 2 / 1.  Same fp32 products, another summation
    order: both within the GEMV tolerance of torch and of each other, for every prologue / epilogue form the decode step uses."""
    from omnitokenizer_amd import _lib
    x, w, bias, res = rnd(B, K, seed=11), rnd(N, K, seed=12, scale=0.05), rnd(N, seed=13), rnd(B, N, seed=14)
    g, beta = rnd(K, seed=15, scale=0.1) + 1.0, rnd(K, seed=16, scale=0.1)
    s = torch.cuda.current_stream().cuda_stream
    xd, wd, bd, rd, gd, betad = (t.cuda() for t in (x, w, bias, res, g, beta))
    tol = 2e-5 * math.sqrt(K / 1536)
    refs = {"plain": F.linear(x, w, bias), "ln_gelu": F.gelu(F.linear(F.layer_norm(x, (K,), g, beta), w, bias)),
            "residual": F.linear(x, w) + res}
    outs = {}
    try:
        for mode in (1, 0):
            _lib.set_option("lm_mfma", mode)
            y = torch.empty(B, N, device="cuda")
            assert lib.omnitok_lm_gemv(_p(xd), _p(wd), _p(bd), None, None, None, _p(y), B, N, K, 0, s) == 0
            outs[mode, "plain"] = y.clone()
            assert lib.omnitok_lm_gemv(_p(xd), _p(wd), _p(bd), None, _p(gd), _p(betad), _p(y), B, N, K, 1, s) == 0
            outs[mode, "ln_gelu"] = y.clone()
            y = rd.clone()
            assert lib.omnitok_lm_gemv(_p(xd), _p(wd), None, _p(y), None, None, _p(y), B, N, K, 0, s) == 0
            outs[mode, "residual"] = y.clone()
    finally:
        _lib.set_option("lm_mfma", 0)
    for form, ref in refs.items():
        for mode in (1, 0):
            assert (outs[mode, form].cpu() - ref).abs().max().item() < tol, (form, mode)
        assert (outs[1, form] - outs[0, form]).abs().max().item() < tol, form
    assert not torch.equal(outs[1, "plain"], outs[0, "plain"])   # the option is live (another summation order)


@pytest.fixture(scope="module")
def gpts():
    cache = {}

    def get(name):
        if name not in cache:
            from omnitokenizer_amd.gpt import GPT
            g, sd, (V, BS, L, H, C) = load_gpt_case(name)
            m = GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C)
            res = m.load_state_dict(sd, strict=True)
            assert not res.missing_keys and not res.unexpected_keys
            cache[name] = (m.cuda().eval(), g, sd, H)
        return cache[name]
    return get


@pytest.mark.parametrize("name", GPT_CASES)
def test_gpt_vs_reference_golden(gpts, name):
    from omnitokenizer_amd import gpt as og
    m, g, sd, H = gpts(name)
    idx, cls, steps = torch.from_numpy(g["idx"]).cuda(), torch.from_numpy(g["cls"]).cuda(), int(g["steps"])
    logits, _ = m(idx)
    err = (logits.cpu() - torch.from_numpy(g["logits"])).abs().max().item()
    assert err < LOGIT_TOL, f"logits differ from the reference by {err:.2e}"
    for use_graph in (False, True):
        greedy = og.sample_with_past(idx[:, :3].clone(), m, steps, temperature=0.9, sample_logits=False, top_k=50,
                                     top_p=0.9, use_graph=use_graph)
        assert np.array_equal(greedy.cpu().numpy(), g["greedy"]), f"greedy sample differs (graph={use_graph})"
        a = og.sample_with_past_cfg(cls.clone(), m, steps, sample_logits=False, top_k=64, top_p=1.0, cfg_ratio=1.5,
                                    class_first=True, use_graph=use_graph)
        b = og.sample_with_past_cfg(cls.clone(), m, steps, sample_logits=False, top_k=64, top_p=0.95, cfg_ratio=0.5,
                                    class_first=False, scale_cfg=True, use_graph=use_graph)
        assert np.array_equal(a.cpu().numpy(), g["cfg_a"]) and np.array_equal(b.cpu().numpy(), g["cfg_b"])
    print(f"{name}: logits err {err:.1e}")


def test_gpt_reference_calling_pattern_and_logits(gpts):
    """The reference's own loop body (gpt.py:334-357) runs unchanged on the drop-in class, and the
    per-step logits match the oracle's."""
    m, g, sd, H = gpts("gpt_hd96")
    x = torch.from_numpy(g["idx"])[:, :4]
    steps = 10
    ref_tok, ref_logits = go.sample_with_past(sd, x, H, steps, sample_logits=False, return_logits=True)
    sample = xc = x.cuda()
    cond_len, past, errs = xc.shape[1], None, []
    for n in range(steps):
        logits, _, present = m.forward_with_past(xc, past=past, past_length=(n + cond_len - 1))
        past = [present] if past is None else past + [present]
        logits = logits[:, -1, :] / 1.0
        errs.append((logits.cpu() - ref_logits[:, n]).abs().max().item())
        _, xc = torch.topk(F.softmax(logits, dim=-1), k=1, dim=-1)
        sample = torch.cat((sample, xc), dim=1)
    assert max(errs) < LOGIT_TOL
    assert torch.equal(sample[:, cond_len:].cpu(), ref_tok)


@pytest.mark.parametrize("name", GPT_CASES)
def test_prefill_equals_steps(gpts, name):
    """The batched prefill and T single-token steps leave the same cache behind and give the same
    logits (different summation orders only), including a prefix that spans several 256-key chunks."""
    m, g, sd, H = gpts(name)
    V, BS = m.vocab_size, m.block_size
    T = min(BS - 2, 40)
    idx = torch.randint(0, V, (3, T), generator=torch.Generator().manual_seed(11)).cuda()
    m.reset_streams(3, T + 4)
    stepped = torch.stack([m.step(idx[:, t].contiguous()) for t in range(T)], 1)
    nxt = torch.randint(0, V, (3,), generator=torch.Generator().manual_seed(12)).cuda()
    after_steps = m.step(nxt)
    m.reset_streams(3, T + 4)
    batched = m.prefill(idx, want_logits=True)
    after_prefill = m.step(nxt)
    assert (batched - stepped).abs().max().item() < LOGIT_TOL
    assert (after_prefill - after_steps).abs().max().item() < LOGIT_TOL
    ref = go.forward(sd, torch.cat([idx.cpu(), nxt.cpu()[:, None]], 1), H)
    assert (batched.cpu() - ref[:, :T]).abs().max().item() < LOGIT_TOL
    assert (after_prefill.cpu() - ref[:, T]).abs().max().item() < LOGIT_TOL


def test_prefill_long_prefix_multi_chunk():
    """600-token prefix (3 attention chunks) on a 1-layer model: prefill + sampling vs the oracle."""
    from omnitokenizer_amd import gpt as og
    V, BS, L, H, C = 256, 700, 1, 4, 256
    sd = go.synth_gpt_state(V, BS, L, H, C, seed=21)
    m = og.GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = torch.randint(0, V, (2, 600), generator=torch.Generator().manual_seed(22))
    new, lg = og.sample_with_past(x.cuda(), m, 6, sample_logits=False, return_logits=True)
    ref, rlg = go.sample_with_past(sd, x, H, 6, sample_logits=False, return_logits=True)
    assert (lg.cpu() - rlg).abs().max().item() < LOGIT_TOL
    assert torch.equal(new.cpu(), ref)


def test_gpt_stochastic_sampling_statistics(gpts):
    """multinomial sampling: tokens are valid, reproducible under a seed, and respect top-k."""
    from omnitokenizer_amd import gpt as og
    m, g, sd, H = gpts("gpt_hd64")
    x = torch.from_numpy(g["idx"])[:, :2].cuda()
    torch.manual_seed(3)
    a, la = og.sample_with_past(x, m, 20, temperature=1.0, top_k=8, top_p=1.0, return_logits=True)
    torch.manual_seed(3)
    b = og.sample_with_past(x, m, 20, temperature=1.0, top_k=8, top_p=1.0)
    assert torch.equal(a, b) and a.min() >= 0 and a.max() < m.vocab_size
    top8 = la.topk(8, dim=-1)[1]
    assert (top8 == a[..., None]).any(-1).all()


def test_gpt_errors(gpts, lib):
    m, g, sd, H = gpts("gpt_hd64")
    with pytest.raises(AssertionError):
        m(torch.zeros(1, m.block_size + 1, dtype=torch.long, device="cuda"))
    with pytest.raises(ValueError):
        m.reset_streams(17, 8)
    from omnitokenizer_amd import gpt as og
    with pytest.raises(ValueError, match="block_size"):
        og.sample_with_past(torch.zeros(1, 3, dtype=torch.long, device="cuda"), m, m.block_size)
    from omnitokenizer_amd._lib import OmnitokLmConfig
    h = ctypes.c_void_p()
    cfg = OmnitokLmConfig(100, 16, 1, 3, 300)
    assert lib.omnitok_lm_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"head_dim" in lib.omnitok_last_error()


def test_interleaved_sequences_are_rejected_not_corrupted(gpts):
    """The engine keeps ONE set of K/V streams: the reference's sample_with_past_cfg pattern (conditional and
    unconditional forward_with_past calls in turn, gpt.py:408-409) would silently overwrite the first
    sequence's cache, so handles of replaced streams raise instead."""
    m, g, sd, H = gpts("gpt_hd64")
    a = torch.tensor([[3, 5]], device="cuda")
    b = torch.tensor([[7]], device="cuda")
    _, _, pa = m.forward_with_past(a, past=None)
    past_a = [pa]
    logits, _, p2 = m.forward_with_past(torch.tensor([[2]], device="cuda"), past=past_a, past_length=2)
    past_a.append(p2)
    _, _, pb = m.forward_with_past(b, past=None)            # a second sequence starts: streams are replaced
    with pytest.raises(RuntimeError, match="sample_with_past_cfg"):
        m.forward_with_past(torch.tensor([[4]], device="cuda"), past=past_a, past_length=3)
    # the new sequence itself keeps working
    m.forward_with_past(torch.tensor([[4]], device="cuda"), past=[pb], past_length=1)


def test_stepping_past_the_cache_is_flagged(gpts):
    m, g, sd, H = gpts("gpt_hd64")
    m.reset_streams(1, 4)
    cap = m._cache_shape[1]                 # the cache only grows (other tests may have sized it)
    tok = torch.tensor([1], device="cuda")
    for _ in range(cap):
        m._pos.zero_()                      # keep the position embedding in range; only the cache length grows
        m.step(tok)
    m.check_overflow()                      # `cap` tokens fit
    m._pos.zero_()
    m.step(tok)                             # one more does not
    with pytest.raises(RuntimeError, match="K/V cache"):
        m.check_overflow()
    m.check_overflow()                      # the flag was cleared


def test_lm_state_dict_missing_keys_raise(gpts):
    from omnitokenizer_amd.gpt import GPT
    g, sd, (V, BS, L, H, C) = load_gpt_case("gpt_hd64")
    m = GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C)
    renamed = {"module." + k: v for k, v in sd.items()}
    with pytest.raises(RuntimeError, match="lacks"):
        m.load_state_dict(renamed)
    with pytest.warns(UserWarning):
        m.load_state_dict(renamed, strict=False)


def test_tokens_to_pixels_chain():
    """The consumer chain of lm_transformer.py:262,430-436: encode() ids condition the LM, sampled ids
    (clamped into the codebook range, :433) go through decode()."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd import gpt as og
    from omnitokenizer_amd.config import OmniTokConfig
    args = make_args(2, resolution=64)
    cfg = OmniTokConfig.from_args(args)
    vq = OmniTokenizer_VQGAN(args)
    vq.load_state_dict(synth.synth_state_dict(cfg, 0), strict=True)
    vq = vq.cuda().eval()
    x = synth.synth_video(1, 5, 64).cuda()
    ids = vq.encode(x, False)                       # [1, 2, 8, 8]
    n_first = 64                                    # first-frame tokens condition the LM (frame prediction)
    V, steps = cfg.n_codes, ids.numel() - n_first
    sd = go.synth_gpt_state(V, ids.numel() + 1, 2, 4, 256, seed=4)
    lm = og.GPT(argparse.Namespace(), V, ids.numel() + 1, n_layer=2, n_head=4, n_embd=256)
    lm.load_state_dict(sd, strict=True)
    lm = lm.cuda().eval()
    cond = ids.reshape(1, -1)[:, :n_first]
    new = og.sample_with_past(cond, lm, steps, sample_logits=False, top_k=100, top_p=0.95)
    ref = go.sample_with_past(sd, cond.cpu(), 4, steps, sample_logits=False, top_k=100, top_p=0.95)
    assert torch.equal(new.cpu(), ref)
    index = torch.clamp(torch.cat([cond, new], 1), min=0, max=V - 1)
    video = vq.decode(index, False)                 # flat video ids, omnitokenizer.py:283-286
    assert tuple(video.shape) == (1, 3, 5, 64, 64) and torch.isfinite(video).all()
    assert torch.equal(video, vq.decode(index.reshape(1, 2, 8, 8), False))


# ---- token selection kernel (csrc/lm_select.hip) ---------------------------------------------------------------
@pytest.mark.parametrize("V,top_k,top_p,temp", [(320, 50, 0.9, 0.9), (9193, 2048, 0.9, 1.0), (9193, 64, 1.0, 0.7),
                                                (300, 300, 0.5, 1.0), (520, 1, 0.3, 1.0), (17385, None, None, 1.0),
                                                (9193, 0, 0.7, 1.3), (8193, 8192, 0.95, 1.0), (17385, 100, 0.9, 1.0),
                                                (16384, 2048, 0.9, 1.0)])
def test_select_kernel_vs_reference_semantics(V, top_k, top_p, temp):
    """greedy = argmax; stochastic = inverse CDF over the survivors of the reference's filter (gpt.py:19-51): for u
    at the midpoints of the oracle's CDF intervals the kernel returns exactly the oracle's token, for random u the
    token agrees unless u sits on an interval edge (fp32 vs fp64 prefix sums)."""
    from omnitokenizer_amd import gpt as og
    B = 6
    lg = rnd(B, V, seed=V + (top_k or 0), scale=3.0)
    lg[0, 10] = lg[0, 11]            # a tie
    lg[1, 5] = lg[1].max() + 4.0     # a dominant token
    d = lg.cuda()
    greedy = og.select_tokens(d, sample_logits=False, top_k=top_k, top_p=top_p, temperature=temp)
    vals = torch.div(lg, temp)
    assert torch.equal(greedy.cpu(), vals.argmax(-1))
    # hand-placed uniforms: patch torch.rand through a generator-free path -- call the C entry directly
    lib = __import__("omnitokenizer_amd")._lib.load()
    out = torch.empty(B, dtype=torch.int64, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")

    def run(u):
        ud = u.float().cuda()
        from omnitokenizer_amd._lib import check
        check(lib.omnitok_lm_select(_p(d), None, B, V, float(temp), 1.0, 0.0, -1 if top_k is None else top_k,
                                    1.0 if top_p is None else float(top_p), 1, _p(ud), _p(out), None, _p(err),
                                    torch.cuda.current_stream().cuda_stream), "lm_select")
        return out.cpu()
    oracles = [go.select_inverse_cdf(vals[b], top_k, top_p, 0.0) for b in range(B)]
    for frac in (0.0, 0.3, 0.77, 0.999):
        u, want = [], []
        for b in range(B):
            _, order, cdf = oracles[b]
            r = min(int(frac * len(order)), len(order) - 1)
            lo = 0.0 if r == 0 else float(cdf[r - 1])
            u.append((lo + float(cdf[r])) / 2)
            want.append(int(order[r]))
        # skip intervals narrower than fp32 can resolve
        got = run(torch.tensor(u, dtype=torch.float64))
        for b in range(B):
            _, order, cdf = oracles[b]
            r = min(int(frac * len(order)), len(order) - 1)
            width = float(cdf[r] - (cdf[r - 1] if r else 0.0))
            if width > 1e-5:
                assert int(got[b]) == want[b], (frac, b, int(got[b]), want[b])
    g = torch.Generator().manual_seed(1)
    agree = total = 0
    for _ in range(20):
        u = torch.rand(B, generator=g, dtype=torch.float64)
        got = run(u)
        for b in range(B):
            tok, order, cdf = go.select_inverse_cdf(vals[b], top_k, top_p, float(u[b].float()))
            edge = (cdf - float(u[b].float())).abs().min().item() < 2e-6
            assert int(got[b]) in set(order.tolist())
            if not edge:
                total += 1
                agree += int(got[b]) == tok
    assert agree == total, f"{total - agree} of {total} draws differ from the inverse-CDF oracle"
    assert int(err.item()) == 0


@pytest.mark.parametrize("top_k", [0, 17000])
def test_select_more_survivors_than_the_sort_buffer(top_k):
    """More than 16384 surviving logits (top_k = 0 keeps everything, top_k > 16384 on a large vocabulary -- the vtokens
    models have V = 16384 + class_cond_dim): without a nucleus cut the draw needs no sort and must stay a draw from the
    filtered softmax (it used to return the argmax silently); with a nucleus cut it raises."""
    from omnitokenizer_amd import gpt as og
    B, V = 4, 20000
    lg = rnd(B, V, seed=77, scale=2.0)
    d = lg.cuda()
    lib = __import__("omnitokenizer_amd")._lib.load()
    out = torch.empty(B, dtype=torch.int64, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    from omnitokenizer_amd._lib import check
    vals = lg.double()
    keep = torch.ones_like(vals, dtype=torch.bool)
    if top_k:
        kth = vals.topk(top_k, dim=-1).values[:, -1:]
        keep = vals >= kth
    pr = torch.where(keep, torch.exp(vals - vals.max(-1, keepdim=True).values), torch.zeros_like(vals))
    cdf = pr.cumsum(-1)  # index order: what a draw without a sort walks
    g = torch.Generator().manual_seed(5)
    agree = total = 0
    seen = set()
    for _ in range(25):
        u = torch.rand(B, generator=g, dtype=torch.float64).float()
        check(lib.omnitok_lm_select(_p(d), None, B, V, 1.0, 1.0, 0.0, top_k, 1.0, 1, _p(u.cuda()), _p(out), None, _p(err),
                                    torch.cuda.current_stream().cuda_stream), "lm_select")
        got = out.cpu()
        for b in range(B):
            tgt = float(u[b]) * float(cdf[b, -1])
            want = int((cdf[b] > tgt).nonzero()[0])
            edge = (cdf[b] - tgt).abs().min().item() < 1e-5 * float(cdf[b, -1])
            assert bool(keep[b, got[b]])
            seen.add(int(got[b]))
            if not edge:
                total += 1
                agree += int(got[b]) == want
    assert agree == total and len(seen) > 50, (agree, total, len(seen))  # a draw, not the argmax
    assert int(err.item()) == 0
    # the same filter with a nucleus cut cannot be evaluated: reported, never silently greedy
    with pytest.raises(NotImplementedError, match="nucleus"):
        og.select_tokens(d, sample_logits=True, top_k=top_k, top_p=0.9)
    # ... and a sampling loop reads the flag once at its end through the shared err tensor
    e2 = torch.zeros(1, dtype=torch.int32, device="cuda")
    og.select_tokens(d, sample_logits=True, top_k=top_k, top_p=0.9, err=e2)
    with pytest.raises(NotImplementedError):
        og.check_select_overflow(e2)


def test_select_kernel_cfg_blend_and_errors():
    from omnitokenizer_amd import gpt as og
    B, V = 3, 1000
    lc, lu = rnd(B, V, seed=1, scale=2.0), rnd(B, V, seed=2, scale=2.0)
    t, temp = 1.5, 0.8
    blend_ref = (1 + t) * (lc / temp) - t * (lu / temp)   # the reference's expression on fp32 tensors (gpt.py:428-431)
    tok, blend = og.select_tokens(lc.cuda(), sample_logits=False, top_k=64, top_p=1.0, temperature=temp,
                                  logits_uncond=lu.cuda(), cfg_t=t, return_logits=True)
    assert torch.equal(blend.cpu(), blend_ref)
    assert torch.equal(tok.cpu(), blend_ref.argmax(-1))
    with pytest.raises(RuntimeError):
        og.select_tokens(lc, sample_logits=False)                       # CPU tensor
    with pytest.raises(NotImplementedError):                           # nucleus over > 16384 survivors
        og.select_tokens(rnd(1, 17385, seed=3).cuda(), sample_logits=True, top_k=0, top_p=0.5)
    torch.manual_seed(5)
    a = og.select_tokens(lc.cuda(), True, 50, 0.9)
    torch.manual_seed(5)
    assert torch.equal(a, og.select_tokens(lc.cuda(), True, 50, 0.9))  # torch's generator governs the draw


# ---- the reference GPT's optional inputs: explicit embeddings, vtokens_pos boxes ----------------------------------
def test_gpt_embeddings_and_vtokens_pos_vs_reference_golden():
    from omnitokenizer_amd import gpt as og
    from omnitokenizer_amd.gpt import GPT
    g, sd, (V, BS, L, H, C) = load_gpt_case("gpt_vtok")
    args = argparse.Namespace(sequence_length=int(g["sequence_length"]), resolution=int(g["resolution"]))
    m = GPT(args, V, BS, n_layer=L, n_head=H, n_embd=C, vtokens_pos=True)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.cuda().eval()
    cbox = [tuple(r) for r in g["cbox"].tolist()]
    tbox = [tuple(r) for r in g["tbox"].tolist()]
    emb, idx36, idx24 = (torch.from_numpy(g[k]).cuda() for k in ("emb", "idx36", "idx24"))
    lg, _ = m(idx36, embeddings=emb, cbox=cbox)
    assert (lg.cpu() - torch.from_numpy(g["logits_emb"])).abs().max().item() < LOGIT_TOL
    lg, _ = m(idx24, cbox=cbox, tbox=tbox)
    assert (lg.cpu() - torch.from_numpy(g["logits_tbox"])).abs().max().item() < LOGIT_TOL
    with pytest.raises(ValueError):
        m(idx24)                                   # a vtokens_pos model needs its boxes (gpt.py:221-225)
    # the reference's KV-cached calling pattern: first call with embeddings, then single-token steps
    first, _, present = m.forward_with_past(idx36[:1, :3], embeddings=emb[:1], cbox=cbox[:1])
    assert (first.cpu() - torch.from_numpy(g["first"])).abs().max().item() < LOGIT_TOL
    past, plen = [present], 5
    for t in range(4):
        lgt, _, present = m.forward_with_past(idx36[:1, 3 + t:4 + t], past=past, past_length=plen, cbox=cbox[:1])
        past.append(present)
        plen += 1
        assert (lgt[:, -1].cpu() - torch.from_numpy(g["step_logits"])[:, t]).abs().max().item() < LOGIT_TOL
    # greedy sampling with boxes: per stream like the reference (its cached path is single-stream), and as one batch
    for use_graph in (False, True):
        one = torch.cat([og.sample_with_past(idx36[b:b + 1, :3].clone(), m, int(g["steps"]), temperature=0.8,
                                             sample_logits=False, top_k=40, top_p=0.9, cbox=cbox[b:b + 1],
                                             use_graph=use_graph) for b in range(2)], 0)
        assert np.array_equal(one.cpu().numpy(), g["greedy"])
        both = og.sample_with_past(idx36[:, :3].clone(), m, int(g["steps"]), temperature=0.8, sample_logits=False,
                                   top_k=40, top_p=0.9, cbox=cbox, use_graph=use_graph)
        assert np.array_equal(both.cpu().numpy(), g["greedy"])
    # explicit embeddings on a model without vtokens_pos: oracle parity
    g2, sd2, (V2, BS2, L2, H2, C2) = load_gpt_case("gpt_hd64")
    m2 = GPT(argparse.Namespace(), V2, BS2, n_layer=L2, n_head=H2, n_embd=C2)
    m2.load_state_dict(sd2, strict=True)
    m2 = m2.cuda().eval()
    e2 = rnd(2, 3, C2, seed=8, scale=0.5)
    idx = torch.from_numpy(g2["idx"])[:, :9]
    ref = go.forward(sd2, idx, H2, embeddings=e2)
    out, _ = m2(idx.cuda(), embeddings=e2.cuda())
    assert (out.cpu() - ref).abs().max().item() < LOGIT_TOL


# ---- Net2NetTransformer mirror (reference lm_transformer.py:19-275) ------------------------------------------------
@pytest.mark.parametrize("starts_with_sos,class_first", [(False, False), (True, False), (True, True)])
def test_net2net_transformer_forward_and_sample(starts_with_sos, class_first):
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    from omnitokenizer_amd.lm_transformer import Net2NetTransformer
    from oracle import omnitok_oracle as orc
    targs = make_args(2, resolution=64)
    cfg = OmniTokConfig.from_args(targs)
    tsd = synth.synth_state_dict(cfg, seed=0)
    tok = OmniTokenizer_VQGAN(targs)
    tok.load_state_dict(tsd, strict=True)
    n_cls, L, H, C = 10, 2, 4, 256
    args = argparse.Namespace(class_cond_dim=n_cls, unconditional=False, vtokens=False, block_size=80, n_layer=L,
                              n_head=H, n_embd=C, vtokens_pos=False, n_unmasked=0, starts_with_sos=starts_with_sos,
                              class_first=class_first)
    net = Net2NetTransformer(args, first_stage_model=tok)
    V = cfg.n_codes + n_cls + (1 if starts_with_sos else 0)
    assert net.transformer.vocab_size == V and net.first_stage_vocab_size == cfg.n_codes
    gsd = go.synth_gpt_state(V, 80, L, H, C, seed=6)
    # the Lightning checkpoint layout: prefixed keys, the tokenizer's off-path keys ride along
    full = {f"transformer.{k}": v for k, v in gsd.items()}
    full.update({f"first_stage_model.{k}": v for k, v in tsd.items()})
    net.load_state_dict(full, strict=True)
    net = net.cuda().eval()
    x = synth.synth_image(2, 64, seed=21)                      # 8 x 8 = 64 latent tokens
    c = torch.tensor([3, 7])
    logits, target = net(x.cuda(), c.cuda())
    # oracle composition of the same pipeline (reference lm_transformer.py:136-192)
    with torch.no_grad():
        z = orc.encode(tsd, x, True, cfg).reshape(2, -1)
    ci = c[:, None]
    if starts_with_sos:
        sos = torch.zeros_like(ci)
        ci, zi = ci + 1, z + n_cls + 1
        cz = torch.cat((ci, sos, zi), 1) if class_first else torch.cat((sos, ci, zi), 1)
        prefix = 1
    else:
        zi = z + n_cls
        cz = torch.cat((ci, zi), 1)
        prefix = 0
    assert torch.equal(target.cpu(), zi)
    ref = go.forward(gsd, cz[:, :-1], H)[:, prefix:]
    assert (logits.cpu() - ref).abs().max().item() < LOGIT_TOL
    # sample(): the reference recomputes the full sequence per token (lm_transformer.py:231-247); greedy is token-equal
    steps = 6
    got = net.sample(zi[:, :0].cuda(), cz[:, :prefix + 1].cuda(), steps, temperature=1.0, sample=False, top_k=20)
    seq = cz[:, :prefix + 1]
    for _ in range(steps):
        nxt = go.forward(gsd, seq, H)[:, -1].argmax(-1, keepdim=True)
        seq = torch.cat((seq, nxt), 1)
    assert torch.equal(got.cpu(), seq[:, prefix + 1:])
    # a non-empty x prefix (half-conditioned completion): the reference returns x[:, c.shape[1]:], i.e. the given prefix
    # followed by the samples (lm_transformer.py:247)
    got2 = net.sample(zi[:, :3].cuda(), cz[:, :prefix + 1].cuda(), steps, temperature=1.0, sample=False, top_k=20)
    seq2 = torch.cat((cz[:, :prefix + 1], zi[:, :3]), 1)
    for _ in range(steps):
        seq2 = torch.cat((seq2, go.forward(gsd, seq2, H)[:, -1].argmax(-1, keepdim=True)), 1)
    assert tuple(got2.shape) == (cz.shape[0], 3 + steps) and torch.equal(got2.cpu(), seq2[:, prefix + 1:])
    # and the sampled ids decode through the tokenizer like transformer_eval.py:68-69
    index = torch.clamp(got - n_cls - (1 if starts_with_sos else 0), min=0, max=net.first_stage_model.n_codes - 1)
    pix = net.first_stage_model.decode(torch.cat((index, index[:, :2].repeat(1, 29)), 1)[:, :64], is_image=True)
    assert tuple(pix.shape) == (2, 3, 64, 64) and torch.isfinite(pix).all()


def test_transformer_eval_call_patterns():
    """The bodies of transformer_eval.py::class_condition_generation (its three branches, :41-69) and
    ::frame_prediction (:106-121) run against the mirrors (Net2NetTransformer + GPT + OmniTokenizer_VQGAN) with the
    reference's own statements; greedy variants are checked token by token against the oracle."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    from omnitokenizer_amd.gpt import sample_with_past, sample_with_past_cfg
    from omnitokenizer_amd.lm_transformer import Net2NetTransformer
    targs = make_args(2, resolution=64)
    cfg = OmniTokConfig.from_args(targs)
    tsd = synth.synth_state_dict(cfg, seed=0)
    tok = OmniTokenizer_VQGAN(targs)
    tok.load_state_dict(tsd, strict=True)
    n_cls, L, H, C, BS = 10, 2, 4, 256, 140
    latent_shape = [2, 8, 8]                      # (5 - 1) // 4 + 1 latent frames of 64 // 8 squared
    for starts_with_sos in (False, True):
        args = argparse.Namespace(class_cond_dim=n_cls, unconditional=False, vtokens=False, block_size=BS, n_layer=L,
                                  n_head=H, n_embd=C, vtokens_pos=False, n_unmasked=0, starts_with_sos=starts_with_sos,
                                  class_first=False)
        gpt = Net2NetTransformer(args, first_stage_model=tok)
        V = cfg.n_codes + n_cls + (1 if starts_with_sos else 0)
        gsd = go.synth_gpt_state(V, BS, L, H, C, seed=6)
        gpt.load_state_dict({f"transformer.{k}": v for k, v in gsd.items()}, strict=True)
        gpt = gpt.cuda().eval()
        n_cond = n_cls + (1 if starts_with_sos else 0)
        steps = int(np.prod(latent_shape[1:]))    # image generation: 64 tokens
        batch_size, class_label = 2, 3
        # ---- class_condition_generation ----
        c_indices = torch.tensor([class_label]).repeat(batch_size, 1).to(gpt.device)
        if not starts_with_sos:
            index_sample = sample_with_past(c_indices, gpt.transformer, steps=steps, sample_logits=True, top_k=100,
                                            callback=None, temperature=1.0, top_p=0.9)
            greedy = sample_with_past(c_indices, gpt.transformer, steps=8, sample_logits=False, top_k=100, top_p=0.9)
            assert torch.equal(greedy.cpu(), go.sample_with_past(gsd, c_indices.cpu(), H, 8, sample_logits=False,
                                                                 top_k=100, top_p=0.9))
        else:
            sos = torch.zeros_like(c_indices)
            ci = torch.cat((sos, c_indices + 1), dim=1)
            index_sample = sample_with_past(ci, gpt.transformer, steps=steps, sample_logits=True, top_k=100,
                                            callback=None, temperature=1.0, top_p=0.9)
            cfg_sample = sample_with_past_cfg(c_indices.clone(), gpt.transformer, steps=steps, sample_logits=True,
                                              top_k=100, callback=None, temperature=1.0, top_p=0.9, cfg_ratio=1.5,
                                              class_first=False, scale_cfg=False)
            assert tuple(cfg_sample.shape) == (batch_size, steps)
            greedy = sample_with_past_cfg(c_indices.clone(), gpt.transformer, steps=8, sample_logits=False, top_k=100,
                                          top_p=0.9, cfg_ratio=1.5)
            assert torch.equal(greedy.cpu(), go.sample_with_past_cfg(gsd, c_indices.cpu(), H, 8, sample_logits=False,
                                                                     top_k=100, top_p=0.9, cfg_ratio=1.5))
        index = torch.clamp(index_sample - n_cond, min=0, max=gpt.first_stage_model.n_codes - 1)
        x_sample = gpt.first_stage_model.decode(index, is_image=True)
        samples = torch.clamp(x_sample + 0.5, 0, 1)
        assert tuple(samples.shape) == (batch_size, 3, 64, 64) and torch.isfinite(samples).all()
    # ---- frame_prediction (unconditional LM over the tokenizer's vocabulary) ----
    uargs = argparse.Namespace(class_cond_dim=None, unconditional=True, vtokens=False, block_size=BS, n_layer=L, n_head=H,
                               n_embd=C, vtokens_pos=False, n_unmasked=0)
    gpt = Net2NetTransformer(uargs, first_stage_model=tok)
    assert gpt.cond_stage_vocab_size == 0 and gpt.transformer.vocab_size == cfg.n_codes
    gpt.load_state_dict({f"transformer.{k}": v for k, v in go.synth_gpt_state(cfg.n_codes, BS, L, H, C, seed=7).items()},
                        strict=True)
    gpt = gpt.cuda().eval()
    input_videos = synth.synth_video(2, 5, 64, seed=33).cuda()
    _, prefix_encodings = gpt.first_stage_model.encode(input_videos, is_image=False, include_embeddings=True)
    prefix_encodings = prefix_encodings[:, :1]                      # (the reference conditions on its first latent frames)
    B, _, Hh, Ww = prefix_encodings.shape
    prefix_encodings = prefix_encodings.view(B, -1)
    steps = int(np.prod(latent_shape))
    index_sample = sample_with_past(prefix_encodings, gpt.transformer, steps=int(steps - 1 * Hh * Ww), sample_logits=True,
                                    top_k=2048, temperature=1.0, top_p=0.9)
    index = torch.clamp(index_sample, min=0, max=gpt.first_stage_model.n_codes - 1)
    index = torch.cat((prefix_encodings, index), dim=1).reshape(B, -1, Hh, Ww)    # "b (t h w) -> b t h w"
    x_sample = gpt.first_stage_model.decode(index, is_image=False)
    assert tuple(x_sample.shape) == (2, 3, 5, 64, 64) and torch.isfinite(x_sample).all()
