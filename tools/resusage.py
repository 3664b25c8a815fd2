#!/usr/bin/env python
"""Compile one HIP source for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line
per kernel: VGPRs / AGPRs / spills / scratch / occupancy.  python tools/resusage.py csrc/gemm_x3.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
       "-Wno-unused-result", "-x", "hip", "-c", src, "-o", "/tmp/_res.o", "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:])
    sys.exit(1)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            pass
        cur = name
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for n, d in rows.items():
    if flt and flt not in n:
        continue
    short = re.sub(r"omnitok::|\(.*\)$|void ", "", n)
    print(f"{short[:90]:90s} vgpr {d.get('VGPRs','?'):>4s} agpr {d.get('AGPRs','?'):>3s} vspill {d.get('VGPRs Spill','?'):>3s} "
          f"sspill {d.get('SGPRs Spill','?'):>3s} scratch {d.get('ScratchSize [bytes/lane]','?'):>4s} occ {d.get('Occupancy [waves/SIMD]','?')}")
