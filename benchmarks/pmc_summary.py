#!/usr/bin/env python
"""Sums rocprofv3 counter_collection CSVs per dispatch and averages over the dispatches of one kernel.

    python benchmarks/pmc_summary.py [--steps N | --last N --calls C] <kernel substring> <dir> [<dir> ...]
      -> JSON {counter: mean per dispatch, "by_grid": {grid size: {counter: mean per dispatch}},
               "per_step": {counter: sum over ALL matching dispatches / N}}     (per_step only with --steps)

--last N --calls C: the traced program made C solve calls in all (set-up seeding included, every call dispatching the
kernel equally often); only the dispatches of the LAST N calls are used and "per_step" is their sum / N.

One step of the headline path may launch the same kernel twice (a small learning pass and the main pass), so the
means are also given per grid size, and per step of the profiled program when the number of steps is named.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarise(kernel, dirs, steps=None, last=None, calls=None):
    out = {}
    by_grid = defaultdict(dict)
    per_step = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(float))
            grid_of = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel not in row.get("Kernel_Name", ""):
                        continue
                    per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                    grid_of[row["Dispatch_Id"]] = row.get("Grid_Size", "?")
            for name, disp in per.items():
                if last and calls:
                    ids = sorted(disp, key=lambda d: int(d))
                    k = max(1, len(ids) // calls)
                    disp = {d: disp[d] for d in ids[-last * k:]}
                    steps = last
                vals = list(disp.values())
                out[name] = sum(vals) / len(vals)
                out[name + "__launches"] = len(vals)
                if steps:
                    per_step[name] = sum(vals) / steps
                groups = defaultdict(list)
                for did, v in disp.items():
                    groups[grid_of[did]].append(v)
                for g, vs in groups.items():
                    by_grid[g][name] = sum(vs) / len(vs)
                    by_grid[g][name + "__launches"] = len(vs)
    if len(by_grid) > 1:
        out["by_grid"] = {g: by_grid[g] for g in sorted(by_grid)}
    if steps:
        out["per_step"] = per_step
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    steps = last = calls = None
    while args and args[0] in ("--steps", "--last", "--calls"):
        if args[0] == "--steps":
            steps = int(args[1])
        elif args[0] == "--last":
            last = int(args[1])
        else:
            calls = int(args[1])
        args = args[2:]
    print(json.dumps(summarise(args[0], args[1:], steps, last, calls), indent=1, sort_keys=True))
