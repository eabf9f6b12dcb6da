#!/usr/bin/env python
"""HBM traffic of one step from the per-kernel PMC summaries of benchmarks/profile.sh (pmc_<kernel>_s03/s10.json):
    python3 benchmarks/pmc_traffic.py <profile dir> > pmc_traffic.json
Counters are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte read requests at
64 bytes, so reads are doubled; WRITE_SIZE is exact."""
import json
import os
import sys

d = sys.argv[1]
out = {"_note": "HBM traffic of one bench.py step (mpqp_screen_kernel<4>, mpqp_gram_kernel<6>, qp_solve_kernel<6,true> incl. its "
                "learning pre-pass, which returns at once with a seeded table; 1M C2 states, certificate table restored to "
                "its set-up state before every step): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                "`python3 benchmarks/qp_launch.py <scale> 6`, last 3 steps (benchmarks/profile.sh, pmc_summary.py --last 3). "
                "KiB counters; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); WRITE_SIZE exact. Algorithmic bytes "
                "per step: 32 MB in (states) + 68 MB out (u0, V, status, active bits)."}
for tag in ("s03", "s10"):
    rec = {}
    f = w = 0.0
    for k in ("screen", "gram", "solve"):
        path = os.path.join(d, "pmc_%s_%s.json" % (k, tag))
        if not os.path.exists(path):
            continue
        ps = json.load(open(path)).get("per_step", {})
        rec[k + "_fetch_size_kib"] = ps.get("FETCH_SIZE")
        rec[k + "_write_size_kib"] = ps.get("WRITE_SIZE")
        f += ps.get("FETCH_SIZE") or 0.0
        w += ps.get("WRITE_SIZE") or 0.0
    rec["fetch_bytes_corrected"] = 2 * f * 1024
    rec["write_bytes"] = w * 1024
    rec["hbm_bytes_per_launch"] = rec["fetch_bytes_corrected"] + rec["write_bytes"]
    rec["algorithmic_bytes_per_launch"] = 100000000
    out[tag] = rec
print(json.dumps(out, indent=1))
