"""tools/pmc_to_json.py: the counter summaries -> profiles/pmc_traffic.json step behind bench.py's static roofline fields
(unit corrections, dispatch-weighted family means, the csrc stamp bench.py compares)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import pmc_to_json  # noqa: E402


def _csv(path, header, rows):
    with open(path, "w") as f:
        f.write(",".join(header) + "\n")
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")


def test_family_means_units_and_stamp(tmp_path):
    ff = '"void omnitok::gemm_pl_kernel<1, false, omnitok::PlCfg<4, 2, 4, 2, 0, 2, 4> >(omnitok::PlParams)"'
    f32 = '"void omnitok::gemm_pl_kernel<0, false, omnitok::PlCfg<4, 2, 4, 2, 0, 2, 4> >(omnitok::PlParams)"'
    fetch, write, mfma = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "m.csv"
    _csv(fetch, ["kernel", "dispatches", "mean_us_under_pmc", "mean_FETCH_SIZE"], [[ff, 16, 1300.0, 1000.0], [f32, 26, 700.0, 500.0],
                                                                                   ["other_kernel", 3, 1.0, 7.0]])
    _csv(write, ["kernel", "dispatches", "mean_us_under_pmc", "mean_WRITE_SIZE"], [[ff, 16, 1300.0, 900.0], [f32, 26, 700.0, 300.0]])
    _csv(mfma, ["kernel", "dispatches", "mean_us_under_pmc", "mean_SQ_VALU_MFMA_BUSY_CYCLES", "mean_GRBM_GUI_ACTIVE",
                "mean_SQ_WAVE_CYCLES", "mean_SQ_WAIT_ANY", "mean_SQ_WAIT_INST_ANY", "mean_SQ_ACTIVE_INST_ANY"],
         [[ff, 16, 1300.0, 128.0 * 500.0, 8.0 * 1000.0, 4000.0, 1000.0, 2000.0, 1200.0]])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), str(fetch), str(write), str(mfma), "rXX"],
                         capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    assert d["stamp"]["csrc_sha256"] == pmc_to_json.csrc_digest() and d["stamp"]["tag"] == "rXX"
    fam = d["gemm_mode_2"]
    # FETCH_SIZE: KB per dispatch, x2 on gfx950; WRITE_SIZE: KB per dispatch
    assert fam["gemm_ff_in"]["read_bytes"] == 1000 * 1024 * 2 and fam["gemm_ff_in"]["write_bytes"] == 900 * 1024
    # busy cycles over (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs
    assert fam["gemm_ff_in"]["mfma_busy_pct"] == round(100.0 * 128.0 * 500.0 / (1000.0 * 1024.0), 1)
    assert fam["gemm_ff_in"]["sq_wait_inst_any_frac_of_wave_cycles"] == 0.5
    # the fp32-epilogue kernel serves FF-out and the fp32 q|k|v launches: never reported as gemm_ff_out
    assert "gemm_ff_out" not in fam and fam["gemm_f32_epilogue_mixed"]["read_bytes"] == 500 * 1024 * 2
    assert "mfma_busy_pct" not in fam["gemm_f32_epilogue_mixed"]
    assert "source" in fam["gemm_ff_in"] and "rXX_pmc_FETCH_SIZE.csv" in fam["gemm_ff_in"]["source"]


def test_digest_changes_with_the_sources(tmp_path):
    d = tmp_path / "omnitokenizer_amd" / "csrc"
    d.mkdir(parents=True)
    (d / "a.hip").write_text("x")
    (d / "b.h").write_text("y")
    (d / "notes.txt").write_text("ignored")
    h1 = pmc_to_json.csrc_digest(str(tmp_path))
    (d / "notes.txt").write_text("still ignored")
    assert pmc_to_json.csrc_digest(str(tmp_path)) == h1
    (d / "b.h").write_text("y2")
    assert pmc_to_json.csrc_digest(str(tmp_path)) != h1
