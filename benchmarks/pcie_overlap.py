#!/usr/bin/env python
"""How much of the host-to-host pipeline's transfers overlap its kernels: from a rocprofv3 --kernel-trace
--memory-copy-trace run of benchmarks/pcie_pipeline.py (last `--steps` steps).
    python3 benchmarks/pcie_overlap.py <trace dir> --steps 12"""
import argparse
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--steps", type=int, required=True)
args = ap.parse_args()


def rows(pattern):
    out = []
    for p in glob.glob(os.path.join(args.dir, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(p)))
    return out


kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows("*kernel_trace.csv")]
cop = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in rows("*memory_copy_trace.csv")]
solve = sorted(k for k in kern if "pympc::" in k[2] and "cuts_mark" not in k[2])
screens = [k for k in solve if "mpqp_screen_kernel" in k[2]][-args.steps:]
t0, t1 = screens[0][0], max(k[1] for k in solve)
blit = [k for k in kern if "copyBuffer" in k[2] and k[0] >= t0 and k[1] <= t1]        # device-to-host copies run as kernels
h2d = [c for c in cop if c[0] >= t0 and c[1] <= t1]
busy = sorted((k[0], k[1]) for k in solve if k[0] >= t0)


def covered(a, b):
    tot = 0
    for s, e in busy:
        lo, hi = max(a, s), min(b, e)
        if hi > lo:
            tot += hi - lo
    return min(tot, b - a)


def summary(items):
    dur = sum(e - s for s, e, _ in items)
    ov = sum(covered(s, e) for s, e, _ in items)
    return dict(count=len(items), total_ms=dur / 1e6, overlapped_with_solve_kernels_ms=ov / 1e6, overlapped_fraction=ov / dur if dur else None)


union = 0
last = t0
for s, e in busy:
    s = max(s, last)
    if e > s:
        union += e - s
        last = e
print(json.dumps({
    "steps": args.steps,
    "window_ms": (t1 - t0) / 1e6,
    "ms_per_step": (t1 - t0) / 1e6 / args.steps,
    "solve_kernels_busy_ms": union / 1e6,
    "uploads_sdma": summary(h2d),
    "downloads_blit_kernels": summary(blit),
    "note": "window = first screening kernel of the last `steps` steps to the end of the last solve kernel; a transfer counts as "
            "overlapped while any kernel of the solve path runs"}, indent=1))
