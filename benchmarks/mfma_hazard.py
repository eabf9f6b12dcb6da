#!/usr/bin/env python
"""Runs the v_mfma_f64_16x16x4_f64 read-after-write probe (pympc_amd/csrc/probes/mfma_f64_hazard.hip) with and without
the wait-state pad round 1 put behind every MFMA chain of csrc/condense.hpp and prints one JSON line per variant."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "pympc_amd", "_lib", "libpympc_probe.so"))
lib.pympc_probe_mfma_f64.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong),
                                     ctypes.POINTER(ctypes.c_int)]
for ks in (1, 2, 4):
    for pad in (1, 0):
        wrong = ctypes.c_longlong(-1); mask = ctypes.c_int(0)
        rc = lib.pympc_probe_mfma_f64(pad, ks, 4096, ctypes.byref(wrong), ctypes.byref(mask))
        print(json.dumps(dict(probe="v_mfma_f64_16x16x4_f64 chain -> immediate store of the accumulator", k_steps=ks, pad=bool(pad),
                              rc=rc, products=4096, wrong_entries=wrong.value, wrong_accumulator_registers=mask.value)), flush=True)
