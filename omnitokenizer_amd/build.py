"""Builds libomnitok.so (hand-written HIP kernels + C-ABI engine) for gfx950 with hipcc.

In-tree build: objects under omnitokenizer_amd/csrc/_obj/, library at
omnitokenizer_amd/lib/libomnitok.so (git-ignored, but it travels to the GPU box with the
snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libomnitok.so")
SOURCES = ["common.cpp", "gemm.hip", "gemm_x3.hip", "gemm_h2.hip", "gemm_pl.hip", "gemm_plt.hip", "norm.hip", "peg.hip", "attn_spatial.hip", "attn_h2.hip", "attn_temporal.hip",
           "vq.hip", "resample.hip", "engine.hip", "engine_build.hip", "engine_run.hip", "lm.hip", "lm_select.hip", "debug.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_x_common.h"), os.path.join(CSRC, "h2_common.h"), os.path.join(CSRC, "gemm_pl.h"), os.path.join(CSRC, "gemm_plt.h"), os.path.join(CSRC, "engine.h"), os.path.join(CSRC, "planes.h"), os.path.join(CSRC, "peg_wide.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_lm.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_debug.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_comm.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-Wno-unused-result"]
# No packed-fp32 VALU instruction (v_pk_{fma,mul,add}_f32) in ANY kernel of the library: the one wrong value this engine ever
# produced was the low half of op_sel-broadcast v_pk_mul/fma_f32 issued beside another workgroup's MFMA stream
# (profiles/r05_temporal_plt.txt; cause not vendor-confirmed).  The target feature is switched off for the device compile -- the
# instruction selector then cannot form them, whatever the vectorisers or hand-written float4 arithmetic ask for (same IEEE
# results: a packed op is two independent fp32 ops) -- and build() scans the ISA it produced (ISA_FORBIDDEN) and refuses to link
# a library that contains one.  The host pass of hipcc prints "not a recognized feature" for it and ignores it.
# Cost at C3: profiles/r06_no_packed_fp32.txt.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# Exception, by measurement: peg.hip.  Its wide-slab kernel holds the 27 stencil weights as register PAIRS for v_pk_fma_f32; without
# the packed form it spills 146 registers (256 VGPRs, 524 B of scratch per lane) on an HBM-bound kernel worth 2 ms of the C3 step.
# Its packed ops are plain pairs -- no op_sel / op_sel_hi broadcast, the only form that ever failed (variant E of the r05 table:
# hand-written plain v_pk_fma_f32 chains, 90 / 90 launches right) -- and the scan below enforces exactly that for this file.
PACKED_PLAIN_OK = {"peg.hip"}
# measurement builds only (e.g. OMNITOK_EXTRA_FLAGS=-DOMNITOK_PL_MEASUREMENT_BUILDS for tools/pl_bench's ablation arms);
# they are ADDED to the flags above, the ISA scan below stays on
FLAGS += os.environ.get("OMNITOK_EXTRA_FLAGS", "").split()
ISA_FORBIDDEN = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b", re.M)
ISA_FORBIDDEN_OPSEL = re.compile(r"^\s*v_pk_(?:fma|mul|add)_f32\b[^\n]*\bop_sel", re.M)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


# per-file flags.  gemm_plt.hip: no SLP vectorisation either (the r05 guard, kept: its register budget was tuned with it);
# OMNITOK_PLT_FLAGS adds flags, it cannot remove the guard
FILE_FLAGS = {"gemm_plt.hip": ["-fno-slp-vectorize", *os.environ.get("OMNITOK_PLT_FLAGS", "").split()]}


def flags_for(src: str) -> list:
    return [*FLAGS, *([] if src in PACKED_PLAIN_OK else NO_PACKED_F32), *FILE_FLAGS.get(src, [])]


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(extra).encode())
    return h.hexdigest()


def scan_isa(asm: str) -> dict:
    """Counts of the instruction classes the build refuses (packed fp32 VALU) and reports (MFMA, scratch) in one device .s file."""
    return {"packed_f32": len(ISA_FORBIDDEN.findall(asm)),
            "packed_f32_op_sel": len(ISA_FORBIDDEN_OPSEL.findall(asm)),
            "mfma": len(re.findall(r"^\s*v_mfma_", asm, re.M)),
            "scratch": len(re.findall(r"^\s*scratch_(?:load|store)", asm, re.M))}


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    stem = src.rsplit(".", 1)[0]
    obj = os.path.join(OBJ, stem + ".o")
    stamp = obj + ".sha"
    isa = obj + ".isa.json"
    flags = flags_for(src)
    dig = _digest([path] + HEADERS, flags)
    if os.path.exists(obj) and os.path.exists(stamp) and os.path.exists(isa) and open(stamp).read() == dig:
        return obj
    if not src.endswith(".hip"):   # host-only C++
        cmd = [_hipcc(), *flags, "-x", "hip", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        report = {"packed_f32": 0, "packed_f32_op_sel": 0, "mfma": 0, "scratch": 0, "device_code": False}
    else:
        # -save-temps=obj leaves the device assembly of the SAME compile next to the object: the scan costs no second compile
        tmp = os.path.join(OBJ, stem + ".tmp")
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        tobj = os.path.join(tmp, stem + ".o")
        cmd = [_hipcc(), *flags, "-save-temps=obj", "-x", "hip", "-c", path, "-o", tobj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        asms = glob.glob(os.path.join(tmp, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))
        if len(asms) != 1:
            raise RuntimeError(f"{src}: expected one device assembly file from -save-temps, found {asms}")
        report = scan_isa(open(asms[0]).read())
        report["device_code"] = True
        bad = report["packed_f32_op_sel"] if src in PACKED_PLAIN_OK else report["packed_f32"]
        if bad:
            raise RuntimeError(f"{src}: {bad} packed-fp32 VALU instructions (v_pk_*_f32"
                               f"{' with op_sel broadcast' if src in PACKED_PLAIN_OK else ''}) in the device ISA -- the library "
                               "must not contain any (build.py NO_PACKED_F32, profiles/r05_temporal_plt.txt)")
        os.replace(tobj, obj)
        shutil.rmtree(tmp, ignore_errors=True)
    with open(isa, "w") as f:
        json.dump(report, f)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def isa_report() -> dict:
    """Per translation unit: what scan_isa found in the objects the library on disk was linked from."""
    out = {}
    for src in SOURCES:
        p = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o.isa.json")
        out[src] = json.load(open(p)) if os.path.exists(p) else None
    return out


def build(verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", LIB)
    return LIB


TORCH_LIB = os.path.join(LIBDIR, "libomnitok_torch.so")


def build_torch_binding(verbose: bool = False) -> str:
    """csrc/torch_binding.cpp -> lib/libomnitok_torch.so: the TORCH_LIBRARY registration of the engine (host C++ only,
    g++ against this PyTorch's headers; the kernels stay in libomnitok.so, found through $ORIGIN)."""
    import torch  # only this step needs it
    from torch.utils import cpp_extension

    src = os.path.join(CSRC, "torch_binding.cpp")
    tdir = os.path.dirname(torch.__file__)
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
             f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    h = hashlib.sha256()
    for p in [src] + HEADERS:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update((" ".join(flags) + torch.__version__).encode())
    stamp = os.path.join(OBJ, "torch_binding.sha")
    os.makedirs(OBJ, exist_ok=True)
    if os.path.exists(TORCH_LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest() \
            and os.path.getmtime(TORCH_LIB) >= os.path.getmtime(LIB):
        return TORCH_LIB
    inc = [os.path.join(os.path.dirname(HERE), "include"), "/opt/rocm/include"] + cpp_extension.include_paths()
    cmd = ["g++", *flags, src, "-o", TORCH_LIB, *[f"-I{i}" for i in inc], f"-L{os.path.join(tdir, 'lib')}", "-ltorch",
           "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", f"-L{LIBDIR}", "-lomnitok", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed for torch_binding.cpp:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(h.hexdigest())
    if verbose:
        print("built", TORCH_LIB)
    return TORCH_LIB


if __name__ == "__main__":
    build(verbose=True)
    build_torch_binding(verbose=True)
