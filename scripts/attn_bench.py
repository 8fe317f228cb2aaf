"""Attention kernel (fa_kernel, fastgen_amd/csrc/wan.hip) at the shapes of the transformer rows, one shape after another so that a
`rocprofv3 --kernel-trace` of this script gives per-shape kernel durations (fg_op_attention allocates and frees its key-split scratch
around the launch: host-side timers would measure that):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/attn -- python3 scripts/attn_bench.py
    python3 scripts/attn_bench.py --parse gpurun_out/attn      # per-shape table from the trace
Shapes: video DiT 1.3B self-attention of chunk k (4680 queries over 1560 * 3 (k + 1) keys, 12 heads x 128), its cross-attention (512 text
keys), DiT-XL/2 (B = 256, 16 heads x 72, 256 tokens)."""
import ctypes
import os
import glob
import sys

SHAPES = [("wan self chunk 0", 1, 12, 128, 4680, 4680), ("wan self chunk 3", 1, 12, 128, 4680, 4 * 4680), ("wan self chunk 6", 1, 12, 128, 4680, 7 * 4680),
          ("wan cross (text)", 1, 12, 128, 4680, 512), ("wan teacher-forcing chunk 6 of B=2", 2, 12, 128, 4680, 7 * 4680),
          ("DiT-XL/2 B=256", 256, 16, 72, 256, 256), ("DiT-XL/2 B=64", 64, 16, 72, 256, 256)]
REPS = 4
if os.environ.get("ATTN_SHAPES"):  # e.g. ATTN_SHAPES=2 : only "wan self chunk 6" (counter passes)
    SHAPES = [SHAPES[int(i)] for i in os.environ["ATTN_SHAPES"].split(",")]

if "--pmc" in sys.argv:  # per-kernel counter averages of a `rocprofv3 --pmc ... --output-format csv` directory
    import csv
    from collections import defaultdict

    d = sys.argv[sys.argv.index("--pmc") + 1]
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(n in r["Kernel_Name"] for n in ("fa_kernel", "fa2_kernel", "fa72_seq_kernel")):
                key = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
                acc[key] += float(r["Counter_Value"])
                cnt[key] += 1
    for key in sorted(acc):
        print(f"{key[0]:42s} {key[1]:28s} {acc[key] / cnt[key]:16.0f}  (n={cnt[key]})")
    sys.exit(0)

if "--parse" in sys.argv:
    import csv

    d = sys.argv[sys.argv.index("--parse") + 1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    fa = [r for r in rows if any(n in r["Kernel_Name"] for n in ("fa_kernel", "fa2_kernel", "fa72_seq_kernel"))]
    cb = [r for r in rows if "fa128_combine" in r["Kernel_Name"]]
    assert len(fa) == REPS * len(SHAPES), (len(fa), len(SHAPES))
    ci = 0
    for i, (name, B, H, hd, Lq, Lkv) in enumerate(SHAPES):
        mine = fa[i * REPS + 1:(i + 1) * REPS]
        us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / len(mine) / 1e3
        # a combine launch directly after an fa launch belongs to it
        extra = 0.0
        for r in mine:
            nxt = [c for c in cb if int(c["Start_Timestamp"]) >= int(r["End_Timestamp"])]
            later_fa = [x for x in fa if int(x["Start_Timestamp"]) > int(r["Start_Timestamp"])]
            if nxt and (not later_fa or int(nxt[0]["Start_Timestamp"]) < int(later_fa[0]["Start_Timestamp"])):
                extra += (int(nxt[0]["End_Timestamp"]) - int(nxt[0]["Start_Timestamp"])) / 1e3
        extra /= len(mine)
        gf = 4.0 * B * H * Lq * Lkv * hd / 1e9
        print(f"{name:38s} fa {us:9.1f} us (+ combine {extra:6.1f})  {gf:8.1f} GFLOP  {gf / (us + extra) * 1e3:7.1f} TFLOP/s  "
              f"{100 * gf / (us + extra) * 1e3 / 2500:5.1f} % of 2.5 PF")
    sys.exit(0)

import torch

from fastgen_amd import _lib
if os.environ.get("FA_TIMING_LIB") == "1":  # scripts/fa_ablate.sh: the FASTGEN_AMD_FA_ABL switches exist only in the timing library
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libfastgen_amd.so", "libfastgen_amd_timing.so")

if os.environ.get("FA_LIB"):  # an experimental build of the library (absolute or repo-relative path)
    _lib.LIB_PATH = os.path.abspath(os.environ["FA_LIB"])

L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
for name, B, H, hd, Lq, Lkv in SHAPES:
    q = torch.randn(B, Lq, H * hd, device="cuda").bfloat16()
    k = torch.randn(B, Lkv, H * hd, device="cuda").bfloat16()
    v = torch.randn(B, Lkv, H * hd, device="cuda").bfloat16()
    out = torch.empty_like(q)
    for _ in range(REPS):
        _lib.check(L.fg_op_attention(p(q), p(k), p(v), p(out), B, H, hd, Lq, Lkv, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    print(name, "done", flush=True)
