"""GPU (-m gpu): kernels of DIFFERENT launches sharing compute units.

Inside one engine every kernel runs alone on its stream.  Two engines on two streams, or the LM consumer's graph replay next to a
tokenizer decode, put workgroups of different kernels on the same CU -- VALU-heavy epilogues beside another launch's MFMA stream,
the condition under which round 5 saw its one wrong value (packed fp32 with op_sel broadcast, profiles/r05_temporal_plt.txt; the
library is built without that instruction class since round 6, omnitokenizer_amd/build.py NO_PACKED_F32).  These tests run such
pairs concurrently from two host threads for >= 200 overlapped steps and compare EVERY output bit for bit with the serial run of
the same call (reference contract: encode / decode are pure functions of their inputs, omnitokenizer.py:247-317).
"""
import argparse
import threading

import numpy as np
import pytest
import torch

from tests.helpers import GoldenCase
from tests.test_oracle_gpt import load_gpt_case

pytestmark = pytest.mark.gpu
import os  # noqa: E402
STEPS = int(os.environ.get("OMNITOK_CORESIDENCY_STEPS", "200"))   # (profiles/r06_no_packed_fp32.txt: one 3000-step run per pair)


def _model(case):
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    m = OmniTokenizer_VQGAN(case.args, attention_mode=case.mode)
    m.load_state_dict(case.sd, strict=True)
    return m.cuda().eval()


class Worker(threading.Thread):
    """Runs `fn()` `steps` times on its own stream; `fn` returns tensors, each compared on the device with the serial reference
    (no host synchronisation inside the loop beyond what the call itself does)."""

    def __init__(self, fn, ref, steps, barrier):
        super().__init__(daemon=True)
        self.fn, self.ref, self.steps, self.barrier = fn, ref, steps, barrier
        self.bad = None
        self.err = None
        self.done = 0

    def run(self):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                bad = torch.zeros((), dtype=torch.int64, device="cuda")
                self.barrier.wait(timeout=60)
                for _ in range(self.steps):
                    outs = self.fn()
                    for o, r in zip(outs, self.ref):
                        # bit comparison: views as integers so that NaN / -0.0 cannot hide a difference
                        bad += (o.contiguous().view(torch.int32 if o.dtype == torch.float32 else o.dtype) !=
                                r.view(torch.int32 if r.dtype == torch.float32 else r.dtype)).sum()
                    self.done += 1
                stream.synchronize()
                self.bad = int(bad)
        except BaseException as e:  # noqa: BLE001
            self.err = e
            try:
                self.barrier.abort()
            except Exception:  # noqa: BLE001
                pass


def run_pair(fa, fb, steps=STEPS):
    ra = [t.clone() for t in fa()]
    rb = [t.clone() for t in fb()]
    # serial determinism first: the comparison below means something only if each call repeats itself
    for f, r in ((fa, ra), (fb, rb)):
        for o, rr in zip(f(), r):
            assert torch.equal(o, rr)
    torch.cuda.synchronize()
    bar = threading.Barrier(2)
    wa, wb = Worker(fa, ra, steps, bar), Worker(fb, rb, steps, bar)
    wa.start()
    wb.start()
    wa.join(600)
    wb.join(600)
    assert not wa.is_alive() and not wb.is_alive(), "concurrent workers hung"
    for w in (wa, wb):
        if w.err is not None:
            raise w.err
    assert wa.done == steps and wb.done == steps
    assert wa.bad == 0 and wb.bad == 0, f"outputs differ from the serial run: {wa.bad} / {wb.bad} elements"


def test_two_engines_on_two_streams_are_bit_identical_to_serial():
    """A 17-frame clip engine (fused temporal stage, spatial attention, plane GEMMs) next to an image engine of another
    architecture (stage-1 / legacy attention: relative-position bias, ALiBi): every encode / decode of 200 overlapped steps equals
    the serial result."""
    ca, cb = GoldenCase("s2_sdpa_r256_vid17"), GoldenCase("s1_legacy_r256_img")
    ma, mb = _model(ca), _model(cb)
    xa, xb = ca.x.cuda(), cb.x.cuda().repeat(4, 1, 1, 1)

    def fa():
        ids, z = ma.encode(xa, False, return_latents=True)
        return ids, z, ma.decode(ids, False)

    def fb():
        ids, z = mb.encode(xb, True, return_latents=True)
        return ids, z, mb.decode(ids, True)
    run_pair(fa, fb)


def test_same_architecture_two_engines_different_sizes():
    """Two engines of the headline architecture, one on 2 clips and one on 8 images: the same kernels at different grid sizes
    co-resident (the small-tile GEMM configurations run several workgroups per CU)."""
    c = GoldenCase("s2_sdpa_r256_vid17")
    ma, mb = _model(c), _model(c)
    xa = torch.cat([c.x, c.x.flip(-1)], 0).cuda()
    from omnitokenizer_amd import synth
    xb = synth.synth_image(8, 256, seed=5).cuda()

    def fa():
        ids = ma.encode(xa, False)
        return ids, ma.decode(ids, False)

    def fb():
        ids = mb.encode(xb, True)
        return ids, mb.decode(ids, True)
    run_pair(fa, fb)


def test_lm_graph_replay_beside_tokenizer_decode():
    """The LM consumer's captured decode step (24 x 5 launches of GEMVs / attention per token, replayed as a HIP graph) beside a
    tokenizer decode: greedy tokens and logits of every sampled sequence, and every decoded clip, equal the serial run."""
    from omnitokenizer_amd import gpt as og
    g, sd, (V, BS, L, H, C) = load_gpt_case("gpt_hd64")
    lm = og.GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C)
    lm.load_state_dict(sd, strict=True)
    lm = lm.cuda().eval()
    idx = torch.from_numpy(g["idx"]).cuda()
    steps = int(g["steps"])
    c = GoldenCase("s2_sdpa_r256_vid17")
    m = _model(c)
    ids = m.encode(c.x.cuda(), False)

    def f_lm():
        tok, lg = og.sample_with_past(idx[:, :3].clone(), lm, steps, temperature=0.9, sample_logits=False, top_k=50, top_p=0.9,
                                      use_graph=True, return_logits=True)
        return tok, lg

    def f_tok():
        return (m.decode(ids, False),)
    tok0 = f_lm()[0]
    assert np.array_equal(tok0.cpu().numpy(), g["greedy"])
    run_pair(f_lm, f_tok)
