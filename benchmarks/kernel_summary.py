#!/usr/bin/env python
"""Steady-state per-kernel durations from a rocprofv3 --kernel-trace CSV of benchmarks/qp_launch.py.  Every step of that
program starts with the certificate table's restore kernel (cuts_mark_kernel), so the last `--launches` dispatches of
that kernel mark the steps; set-up (prepare, seeding) and the first `--skip-launches` steps (cold caches, first touch of
the code object) are left out, so that the sum over a step reproduces bench.py's ms_per_step.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 benchmarks/qp_launch.py 0.3 13
    python3 benchmarks/kernel_summary.py DIR --launches 13 --skip-launches 3
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
rows = []
for p in glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
marks = [int(r["Start_Timestamp"]) for r in rows if "cuts_mark_kernel" in r["Kernel_Name"]][-args.launches:]
assert len(marks) == args.launches, "fewer cuts_mark_kernel dispatches than launches"
starts = marks[args.skip_launches:]
n = len(starts)
per = defaultdict(list)
span = []
for i, t0 in enumerate(starts):
    t1 = starts[i + 1] if i + 1 < n else None
    step = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and (t1 is None or int(r["Start_Timestamp"]) < t1)
            and ("pympc" in r["Kernel_Name"] or "cuts_mark" in r["Kernel_Name"])]
    for r in step:
        per[short(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    span.append(max(int(r["End_Timestamp"]) for r in step) - t0)
out = {}
tot = 0.0
for name, d in sorted(per.items()):
    out[name] = dict(dispatches_per_step=len(d) / n, avg_us_per_step=sum(d) / n / 1e3, max_us=max(d) / 1e3, min_us=min(d) / 1e3)
    tot += sum(d) / n / 1e6
out["sum_of_kernels_ms_per_step"] = tot
out["first_start_to_last_end_ms_per_step"] = sum(span) / n / 1e6
out["steps_averaged"] = n
print(json.dumps(out, indent=1))
