"""CPU: no translation unit of libomnitok.so contains a packed-fp32 VALU instruction (build.py NO_PACKED_F32 + its ISA scan; peg.hip:
plain pairs only), and the ISA of the fused temporal stage's kernel (csrc/gemm_plt.h) keeps the two properties its correctness and its speed were
measured with (profiles/r05_temporal_plt.txt): no packed-fp32 VALU instruction and no scratch access in either instantiation, at
256 registers per wave.  The kernel is compiled to assembly with exactly build.py's flags for its translation unit."""
import os
import re
import subprocess

import pytest

from omnitokenizer_amd import build as b


@pytest.fixture(scope="module")
def plt_asm(tmp_path_factory):
    try:
        hipcc = b._hipcc()
    except RuntimeError:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("plt") / "gemm_plt.s"
    src = os.path.join(b.CSRC, "gemm_plt.hip")
    cmd = [hipcc, *b.flags_for("gemm_plt.hip"), "--offload-device-only", "-S", "-x", "hip", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def kernel_body(asm, epi):
    name = f"_ZN7omnitok15gemm_plt_kernelILi{epi}EEEvNS_8PlParamsE"
    m = re.search(rf"^{name}:[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M)
    assert m, f"{name} not found"
    return m.group(1)


@pytest.mark.parametrize("epi", [6, 7])
def test_gemm_plt_has_no_packed_fp32_and_no_scratch(plt_asm, epi):
    body = kernel_body(plt_asm, epi)
    mfma = len(re.findall(r"^\s*v_mfma_f32_32x32x16_f16", body, re.M))
    assert mfma == 30, mfma   # one K step: 2 x 5 accumulator blocks x 3 plane products, and no second copy of the loop
    packed = re.findall(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b", body, re.M)
    assert not packed, f"{len(packed)} packed-fp32 instructions (SLP vectorisation must stay off for gemm_plt.hip)"
    scratch = re.findall(r"^\s*(scratch_\w+|buffer_(?:load|store)\w* .*offen.*scratch)", body, re.M)
    assert not scratch, f"{len(scratch)} scratch accesses: the kernel spills"
    assert "-fno-slp-vectorize" in b.flags_for("gemm_plt.hip")


def test_gemm_plt_register_budget(plt_asm):
    for epi in (6, 7):
        meta = re.search(rf"\.name:\s+_ZN7omnitok15gemm_plt_kernelILi{epi}EEEvNS_8PlParamsE\n(.*?)\.wavefront_size", plt_asm, re.S)
        assert meta, epi
        fields = dict(re.findall(r"\.(\w+):\s+(\d+)", meta.group(1)))
        assert int(fields["vgpr_count"]) <= 256 and int(fields["vgpr_spill_count"]) == 0 and int(fields["private_segment_fixed_size"]) == 0, fields


def test_library_has_no_packed_fp32_instruction():
    """Every object libomnitok.so is linked from was scanned at build time (build.py scan_isa on the device assembly of the same
    compile): no v_pk_{fma,mul,add}_f32 anywhere, except plain (no op_sel) pairs in peg.hip; the guard flags are on every TU."""
    try:
        b._hipcc()
    except RuntimeError:
        pytest.skip("hipcc not available")
    b.build()
    rep = b.isa_report()
    assert set(rep) == set(b.SOURCES)
    for src, r in rep.items():
        assert r is not None, src
        if src in b.PACKED_PLAIN_OK:
            assert r["packed_f32_op_sel"] == 0, (src, r)
        else:
            assert r["packed_f32"] == 0, (src, r)
            assert all(f in b.flags_for(src) for f in b.NO_PACKED_F32), src
    assert "-fno-slp-vectorize" in b.flags_for("gemm_plt.hip")
    assert b.PACKED_PLAIN_OK == {"peg.hip"}


def test_scan_isa_catches_the_instruction_class():
    asm = "\tv_pk_fma_f32 v[86:87], v[188:189], v[44:45], v[32:33] op_sel:[0,1,0]\n\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5]\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]\n"
    r = b.scan_isa(asm)
    assert r["packed_f32"] == 2 and r["packed_f32_op_sel"] == 1 and r["mfma"] == 1


def test_plt_env_flags_cannot_remove_the_guard(monkeypatch):
    import importlib
    monkeypatch.setenv("OMNITOK_PLT_FLAGS", "-DFOO")
    b2 = importlib.reload(b)
    try:
        f = b2.flags_for("gemm_plt.hip")
        assert "-fno-slp-vectorize" in f and "-DFOO" in f and "-packed-fp32-ops" in f
    finally:
        monkeypatch.delenv("OMNITOK_PLT_FLAGS")
        importlib.reload(b)
