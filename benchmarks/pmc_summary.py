#!/usr/bin/env python
"""Sums rocprofv3 counter_collection CSVs per dispatch and averages over the dispatches of one kernel.

    python benchmarks/pmc_summary.py <kernel substring> <dir> [<dir> ...]  -> JSON {counter: mean per launch}
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarise(kernel, dirs):
    out = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(float))
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel not in row.get("Kernel_Name", ""):
                        continue
                    per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for name, disp in per.items():
                vals = list(disp.values())
                out[name] = sum(vals) / len(vals)
                out[name + "__launches"] = len(vals)
    return out


if __name__ == "__main__":
    print(json.dumps(summarise(sys.argv[1], sys.argv[2:]), indent=1, sort_keys=True))
