"""The C++-registered operators of the path (csrc/torch_binding.cpp): the engine as a torch::CustomClassHolder, tensor-only
schemas, torch.export.  CPU part: library loads, shapes, tracing, loud failure off the GPU.  GPU part: the operators give
the module's own encode()/decode() results bit for bit, and an exported program runs."""
import pytest
import torch

from omnitokenizer_amd import make_args, synth, torch_engine
from omnitokenizer_amd.config import OmniTokConfig
from omnitokenizer_amd.vqgan import OmniTokenizer_VQGAN


@pytest.fixture(scope="module")
def te():
    torch_engine.load()
    return torch_engine


def _cpu_engine(te, **over):
    m = OmniTokenizer_VQGAN(make_args(2, **over))
    ints, enc, dec = te.native_config_dict(m)
    return torch.classes.omnitok.Engine(ints, enc, dec), (ints, enc, dec)


def test_engine_class_and_shapes_need_no_gpu(te):
    e, (ints, enc, dec) = _cpu_engine(te)
    assert e.encode_shape(17, 256, 256) == [5, 32, 32] and e.decode_shape(5, 32, 32) == [17, 256, 256]
    assert e.encode_shape(1, 128, 128) == [1, 16, 16]
    assert len(e.missing()) > 100  # nothing set yet: every tensor of the path is listed
    assert e.config()["n_codes"] == ints["n_codes"] and e.blocks() == [enc, dec]
    with pytest.raises(RuntimeError, match="unknown configuration key"):
        torch.classes.omnitok.Engine({**ints, "n_code": 1}, enc, dec)
    with pytest.raises(RuntimeError):
        e.encode_shape(17, 250, 256)  # not a multiple of the patch size: the path rejects it


def test_engine_pickles_as_its_configuration(te):
    import io
    e, (ints, enc, dec) = _cpu_engine(te)
    buf = io.BytesIO()
    torch.save(e, buf)
    buf.seek(0)
    e2 = torch.load(buf, weights_only=False)
    assert dict(e2.config()) == ints and e2.blocks() == [enc, dec]
    assert e2.encode_shape(17, 256, 256) == [5, 32, 32] and len(e2.missing()) == len(e.missing())


def test_operators_fail_loudly_off_the_gpu(te):
    e, _ = _cpu_engine(te)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        torch.ops.omnitok.engine_encode(e, torch.zeros(1, 3, 1, 64, 64))
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        torch.ops.omnitok.engine_decode(e, torch.zeros(1, 1, 8, 8, dtype=torch.int64))


def test_torch_export_captures_encode_and_decode(te):
    e, _ = _cpu_engine(te)
    ep = torch.export.export(te.EngineModule(e), (torch.empty(2, 3, 17, 256, 256, device="meta"),), strict=False)
    calls = [n for n in ep.graph.nodes if n.op == "call_function"]
    assert [str(n.target) for n in calls] == ["omnitok.engine_encode.default", "omnitok.engine_decode.default"]
    assert tuple(calls[0].meta["val"].shape) == (2, 5, 32, 32) and calls[0].meta["val"].dtype == torch.int64
    assert tuple(calls[1].meta["val"].shape) == (2, 3, 17, 256, 256) and calls[1].meta["val"].dtype == torch.float32
    # image call of the same program family
    ep2 = torch.export.export(te.EngineModule(e), (torch.empty(4, 3, 128, 128, device="meta"),), strict=False)
    vals = [n.meta["val"] for n in ep2.graph.nodes if n.op == "call_function"]
    assert tuple(vals[0].shape) == (4, 1, 16, 16) and tuple(vals[1].shape) == (4, 3, 1, 128, 128)


@pytest.mark.gpu
def test_operators_match_the_module_bitwise_and_exported_program_runs(te):
    args = make_args(2, resolution=64)
    cfg = OmniTokConfig.from_args(args)
    model = OmniTokenizer_VQGAN(args)
    model.load_state_dict(synth.synth_state_dict(cfg, seed=3), strict=True)
    model = model.cuda().eval()
    e = te.engine_from_module(model)
    assert e.missing() == []
    x = synth.synth_video(2, 9, 64, seed=4).cuda()
    ids_ref = model.encode(x, is_image=False)
    ids = torch.ops.omnitok.engine_encode(e, x)
    assert ids.dtype == torch.int64 and torch.equal(ids, ids_ref.reshape(ids.shape))
    ids2, emb, z = torch.ops.omnitok.engine_encode_full(e, x)
    emb_ref, ids_ref2 = model.encode(x, is_image=False, include_embeddings=True)
    assert torch.equal(ids2, ids) and torch.equal(ids_ref2, ids) and torch.equal(emb, emb_ref)
    assert z.shape == (*ids.shape, cfg.codebook_dim)
    pix = torch.ops.omnitok.engine_decode(e, ids)
    assert torch.equal(pix, model.decode(ids_ref, is_image=False).reshape(pix.shape))
    # ids outside the codebook: IndexError-like failure, as F.embedding in the reference
    bad = ids.clone()
    bad[0, 0, 0, 0] = cfg.n_codes
    with pytest.raises(RuntimeError, match="id range|out of range|invalid"):
        torch.ops.omnitok.engine_decode(e, bad)
    # an exported program with the engine as a constant
    ep = torch.export.export(te.EngineModule(e), (x,), strict=False)
    out, out_ids = ep.module()(x)
    assert torch.equal(out_ids, ids) and torch.equal(out, pix)
    # image path
    xi = synth.synth_image(3, 64, seed=5).cuda()
    ids_i = torch.ops.omnitok.engine_encode(e, xi)
    assert tuple(ids_i.shape) == (3, 1, 8, 8) and torch.equal(ids_i, model.encode(xi, is_image=True).reshape(ids_i.shape))
    pix_i = torch.ops.omnitok.engine_decode(e, ids_i)
    assert tuple(pix_i.shape) == (3, 3, 1, 64, 64)
    assert torch.equal(pix_i[:, :, 0], model.decode(ids_i, is_image=True))
