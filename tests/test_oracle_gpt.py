"""CPU: pins the LM oracle (oracle/gpt_oracle.py) against the committed outputs of the reference's
own GPT class (tests/golden/gpt_*.npz) and, in the build container, against the live class."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import gpt_oracle as go
from oracle import ref_harness as rh
from tests.helpers import GOLDEN

GPT_CASES = ["gpt_hd64", "gpt_hd96", "gpt_hd128"]


def load_gpt_case(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    V, BS, L, H, C = (int(g[k]) for k in ("vocab", "block_size", "n_layer", "n_head", "n_embd"))
    sd = go.synth_gpt_state(V, BS, L, H, C, seed=int(g["weight_seed"]))
    crc = 0
    for k in sd:
        crc = zlib.crc32(sd[k].numpy().tobytes(), crc)
    assert crc == int(g["state_crc"]), "synthetic GPT weights drifted from the golden fixtures"
    return g, sd, (V, BS, L, H, C)


@pytest.mark.parametrize("name", GPT_CASES)
def test_gpt_oracle_matches_reference_golden(name):
    g, sd, (V, BS, L, H, C) = load_gpt_case(name)
    idx, cls, steps = torch.from_numpy(g["idx"]), torch.from_numpy(g["cls"]), int(g["steps"])
    with torch.no_grad():
        logits = go.forward(sd, idx, H)
        assert (logits - torch.from_numpy(g["logits"])).abs().max().item() < 2e-5
        # the KV-cached step reproduces the full forward
        lg, cache = go.forward_with_past(sd, idx[:, :5], H, None)
        l2, cache = go.forward_with_past(sd, idx[:, 5:6], H, cache, position=5)
        assert (l2[:, 0] - logits[:, 5]).abs().max().item() < 2e-5
        greedy = go.sample_with_past(sd, idx[:, :3], H, steps, temperature=0.9, sample_logits=False, top_k=50, top_p=0.9)
        assert np.array_equal(greedy.numpy(), g["greedy"])
        a = go.sample_with_past_cfg(sd, cls, H, steps, sample_logits=False, top_k=64, top_p=1.0, cfg_ratio=1.5,
                                    class_first=True)
        b = go.sample_with_past_cfg(sd, cls, H, steps, sample_logits=False, top_k=64, top_p=0.95, cfg_ratio=0.5,
                                    class_first=False, scale_cfg=True)
        assert np.array_equal(a.numpy(), g["cfg_a"]) and np.array_equal(b.numpy(), g["cfg_b"])


@pytest.mark.skipif(not rh.reference_available(), reason="needs /root/reference (build container only)")
def test_gpt_oracle_matches_live_reference():
    import argparse
    import importlib
    rh.install_stubs()
    gpt = importlib.import_module("OmniTokenizer.modules.gpt")
    V, BS, L, H, C = 200, 32, 2, 4, 256
    sd = go.synth_gpt_state(V, BS, L, H, C, seed=9)
    m = gpt.GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C).eval()
    m.load_state_dict(sd, strict=False)
    idx = torch.randint(0, V, (3, 12), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref, _ = m(idx)
        assert (go.forward(sd, idx, H) - ref).abs().max().item() < 2e-5
        lg = torch.randn(3, V, generator=torch.Generator().manual_seed(2))
        assert torch.equal(go.top_k_top_p_filtering(lg, 20, 0.8), gpt.top_k_top_p_filtering(lg.clone(), 20, 0.8))


def test_product_gpt_state_dict_and_cpu_refusal():
    import argparse
    from omnitokenizer_amd.gpt import GPT
    _, sd, (V, BS, L, H, C) = load_gpt_case("gpt_hd64")
    m = GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C).eval()
    sd2 = dict(sd)
    sd2["blocks.0.attn.mask"] = torch.ones(1, 1, BS, BS)  # reference buffer, dropped
    res = m.load_state_dict(sd2, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert set(m.state_dict()) == set(sd)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C, vtokens_pos=True)


@pytest.mark.parametrize("top_k,top_p", [(50, 0.9), (2048, 0.9), (64, 1.0), (300, 0.5), (1, 0.3)])
def test_product_token_selection_matches_reference_semantics(top_k, top_p):
    """omnitokenizer_amd.gpt's filtering (pure torch, CPU-runnable) == the oracle's restatement of
    reference gpt.py:19-51, and its single-sort sampling branch draws from exactly that distribution."""
    import torch.nn.functional as F
    from omnitokenizer_amd import gpt as og
    g = torch.Generator().manual_seed(top_k)
    lg = torch.randn(4, 300, generator=g) * 3
    lg[0, 10] = lg[0, 11]
    ref = go.top_k_top_p_filtering(lg, top_k, top_p)
    assert torch.equal(og.top_k_top_p_filtering(lg.clone(), top_k=top_k, top_p=top_p), ref)
    # greedy branch: same token as the reference's topk(softmax(filtered))
    assert torch.equal(og._select(lg.clone(), False, top_k, top_p), torch.topk(F.softmax(ref, -1), 1)[1])
    # stochastic branch: empirical support is inside the reference's support, frequencies follow its probs
    probs = F.softmax(ref, -1)
    torch.manual_seed(0)
    draws = torch.cat([og._select(lg.clone(), True, top_k, top_p) for _ in range(400)], 1)  # [4, 400]
    assert (probs.gather(1, draws) > 0).all()
    if top_k > 1:
        top = probs.argmax(-1)
        freq = (draws == top[:, None]).float().mean(1)
        assert (freq - probs.gather(1, top[:, None])[:, 0]).abs().max().item() < 0.12
