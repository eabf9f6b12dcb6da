#!/usr/bin/env python
"""Steady-state per-kernel durations from a rocprofv3 --kernel-trace CSV: the first `--skip` dispatches of every kernel
(cold caches, first-touch of the code object) are left out, so that the sum over a step reproduces bench.py's ms_per_step.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 benchmarks/qp_launch.py 0.3 13
    python3 benchmarks/kernel_summary.py DIR --skip-launches 3 --launches 13
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--launches", type=int, required=True, help="solve_batch calls the traced command made")
ap.add_argument("--skip-launches", type=int, default=3, help="leading calls to leave out")
args = ap.parse_args()
paths = glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if name.startswith("pympc::") or "cuts_mark" in name:
        per[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = {}
step_ms = 0.0
for name, v in sorted(per.items()):
    if len(v) % args.launches:
        continue            # set-up kernels (prepare, seeding): not part of a step
    k = len(v) // args.launches
    d = [e - s for s, e in v[k * args.skip_launches:]]
    n = args.launches - args.skip_launches
    out[name] = dict(dispatches_per_step=k, avg_us_per_step=sum(d) / n / 1e3, max_us=max(d) / 1e3, min_us=min(d) / 1e3)
    step_ms += sum(d) / n / 1e6
out["sum_ms_per_step"] = step_ms
print(json.dumps(out, indent=1))
