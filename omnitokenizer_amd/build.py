"""Builds libomnitok.so (hand-written HIP kernels + C-ABI engine) for gfx950 with hipcc.

In-tree build: objects under omnitokenizer_amd/csrc/_obj/, library at
omnitokenizer_amd/lib/libomnitok.so (git-ignored, but it travels to the GPU box with the
snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libomnitok.so")
SOURCES = ["common.cpp", "gemm.hip", "gemm_x3.hip", "gemm_h2.hip", "gemm_pl.hip", "norm.hip", "peg.hip", "attn_spatial.hip", "attn_h2.hip", "attn_temporal.hip",
           "vq.hip", "resample.hip", "engine.hip", "engine_build.hip", "engine_run.hip", "lm.hip", "lm_select.hip", "debug.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_x_common.h"), os.path.join(CSRC, "h2_common.h"), os.path.join(CSRC, "gemm_pl.h"), os.path.join(CSRC, "engine.h"), os.path.join(CSRC, "planes.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_lm.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-Wno-unused-result"]
# measurement builds only (e.g. OMNITOK_EXTRA_FLAGS=-DOMNITOK_PL_MEASUREMENT_BUILDS for tools/pl_bench's ablation arms)
FLAGS += os.environ.get("OMNITOK_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
    stamp = obj + ".sha"
    dig = _digest([path] + HEADERS)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc(), *FLAGS, "-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


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


if __name__ == "__main__":
    build(verbose=True)
