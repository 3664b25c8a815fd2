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
    vt = (int(g["sequence_length"]), int(g["resolution"])) if "sequence_length" in g.files else None
    sd = go.synth_gpt_state(V, BS, L, H, C, seed=int(g["weight_seed"]), vtokens_pos_shape=vt)
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


def test_gpt_oracle_optional_inputs_match_reference_golden():
    """explicit embeddings prepended + vtokens_pos boxes (reference gpt.py:207-258), pinned by gpt_vtok.npz."""
    g, sd, (V, BS, L, H, C) = load_gpt_case("gpt_vtok")
    cbox = [tuple(r) for r in g["cbox"].tolist()]
    tbox = [tuple(r) for r in g["tbox"].tolist()]
    emb, idx36, idx24 = (torch.from_numpy(g[k]) for k in ("emb", "idx36", "idx24"))
    with torch.no_grad():
        assert (go.forward(sd, idx36, H, embeddings=emb, cbox=cbox) - torch.from_numpy(g["logits_emb"])).abs().max() < 2e-5
        assert (go.forward(sd, idx24, H, cbox=cbox, tbox=tbox) - torch.from_numpy(g["logits_tbox"])).abs().max() < 2e-5
        first, cache = go.forward_with_past(sd, idx36[:1, :3], H, None, embeddings=emb[:1], cbox=cbox[:1])
        assert (first - torch.from_numpy(g["first"])).abs().max() < 2e-5
        for t in range(4):
            lg, cache = go.forward_with_past(sd, idx36[:1, 3 + t:4 + t], H, cache, position=5 + t, cbox=cbox[:1])
            assert (lg[:, 0] - torch.from_numpy(g["step_logits"])[:, t]).abs().max() < 2e-5
        greedy = torch.cat([go.sample_with_past(sd, idx36[b:b + 1, :3], H, int(g["steps"]), temperature=0.8,
                                                sample_logits=False, top_k=40, top_p=0.9, cbox=cbox[b:b + 1])
                            for b in range(2)], 0)
        assert np.array_equal(greedy.numpy(), g["greedy"])


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
    # n_unmasked only edits the reference's unused mask buffer (gpt.py:98-100 vs the SDPA branch :122-126): accepted
    assert GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C, n_unmasked=2).n_unmasked == 2
    # vtokens_pos: the reference's extra parameter, same key and shape (gpt.py:183-184)
    mv = GPT(argparse.Namespace(sequence_length=3, resolution=6), V, BS, n_layer=L, n_head=H, n_embd=C, vtokens_pos=True)
    assert tuple(mv.state_dict()["vtokens_pos_emb"].shape) == (1, 3, 6, 6, C)
    assert set(mv.state_dict()) == set(sd) | {"vtokens_pos_emb"}


@pytest.mark.parametrize("top_k,top_p", [(50, 0.9), (2048, 0.9), (64, 1.0), (300, 0.5), (1, 0.3), (None, None), (0, 0.7)])
def test_select_inverse_cdf_draws_from_the_reference_distribution(top_k, top_p):
    """The inverse-CDF restatement the selection kernel is tested against (oracle select_inverse_cdf) assigns every
    survivor of the reference's filter (gpt.py:19-51) an interval of u of exactly its softmax probability."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7 if top_k is None else top_k)
    lg = torch.randn(300, generator=g) * 3
    lg[10] = lg[11]
    ref = lg.clone()[None]
    if top_k is not None:
        ref = go.top_k_top_p_filtering(ref, top_k, 1.0 if top_p is None else top_p)
    probs = F.softmax(ref.double(), -1)[0]
    tok, order, cdf = go.select_inverse_cdf(lg, top_k, top_p, 0.0)
    assert tok == int(order[0])
    if top_k is not None:
        assert tok == int(probs.argmax())
    assert set(order.tolist()) == set(torch.nonzero(probs > 0)[:, 0].tolist())
    widths = torch.diff(cdf, prepend=torch.zeros(1, dtype=torch.float64))
    assert (widths - probs[order]).abs().max().item() < 1e-12
    for r in (0, len(order) // 2, len(order) - 1):   # the midpoint of an interval selects its token
        mid = float(cdf[r] - widths[r] / 2)
        assert go.select_inverse_cdf(lg, top_k, top_p, mid)[0] == int(order[r])
