#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
    python scripts/kernel_resources.py fastgen_amd/csrc/conv.hip [substring filter]"""
import re
import subprocess
import sys

src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang",
       "-packed-fp32-ops", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "")
    if flt in n:
        print(f'{r.get("VGPRs","?"):>4} vgpr {r.get("AGPRs","?"):>4} agpr {r.get("VGPRs Spill","?"):>4} spill {r.get("ScratchSize [bytes/lane]","?"):>5} scratch '
              f'{r.get("LDS Size [bytes/block]","?"):>6} lds  occ {r.get("Occupancy [waves/SIMD]","?")}  {n[:150]}')
