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
SOURCES = ["common.cpp", "gemm.hip", "gemm_x3.hip", "gemm_h2.hip", "gemm_pl.hip", "gemm_plt.hip", "norm.hip", "peg.hip", "attn_spatial.hip", "attn_h2.hip", "attn_temporal.hip",
           "vq.hip", "resample.hip", "engine.hip", "engine_build.hip", "engine_run.hip", "lm.hip", "lm_select.hip", "debug.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_x_common.h"), os.path.join(CSRC, "h2_common.h"), os.path.join(CSRC, "gemm_pl.h"), os.path.join(CSRC, "gemm_plt.h"), os.path.join(CSRC, "engine.h"), os.path.join(CSRC, "planes.h"), os.path.join(CSRC, "peg_wide.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_lm.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_debug.h"),
           os.path.join(os.path.dirname(HERE), "include", "omnitok_comm.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-Wno-unused-result"]
# measurement builds only (e.g. OMNITOK_EXTRA_FLAGS=-DOMNITOK_PL_MEASUREMENT_BUILDS for tools/pl_bench's ablation arms)
FLAGS += os.environ.get("OMNITOK_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


# per-file flags.  gemm_plt.hip: no SLP vectorisation -- its epilogues run beside another workgroup's MFMAs on the same SIMD, and the
# v_pk_*_f32 instructions SLP forms there were the one place a wrong value was ever observed (profiles/r05_temporal_plt.txt)
FILE_FLAGS = {"gemm_plt.hip": os.environ.get("OMNITOK_PLT_FLAGS", "-fno-slp-vectorize").split()}


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join([*FLAGS, *extra]).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
    stamp = obj + ".sha"
    extra = FILE_FLAGS.get(src, [])
    dig = _digest([path] + HEADERS, extra)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, "-x", "hip", "-c", path, "-o", obj]
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
