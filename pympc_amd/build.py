"""Builds the in-tree HIP library (gfx950 only) with hipcc.

    python -m pympc_amd.build [--force] [--verbose]

The shared object lands in pympc_amd/_lib/libpympc_hip.so; it is git-ignored but travels with the tree to the
GPU box.  hipcc cross-compiles here without a GPU.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB_PATH = os.environ.get("PYMPC_HIP_LIB") or os.path.join(LIB_DIR, "libpympc_hip.so")   # (override: debug builds)
SOURCES = ["pympc_hip.hip"]
HEADERS = ["ws_layout.hpp", "qp_wave.hpp", "qp_screen.hpp", "qp_gram.hpp", "nnls_raw.hpp", "nnls_block.hpp", "qp_block.hpp", "qp_prepare.hpp", "point_location.hpp", "lp_block.hpp", "lp_tile.hpp", "invqr_wave.hpp", "condense.hpp",
           os.path.join("..", "..", "include", "pympc_hip.h")]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        path = os.path.join(CSRC, f)
        if os.path.exists(path) and os.path.getmtime(path) > t:
            return True
    return False


def build(force=False, verbose=False):
    """Compile pympc_amd/csrc/*.hip for gfx950 into pympc_amd/_lib/libpympc_hip.so; returns the path."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


PROBE_PATH = os.path.join(LIB_DIR, "libpympc_probe.so")


def build_probes(force=False):
    """Small stand-alone probes (csrc/probes/*.hip) in one library: hardware / compiler facts the kernels rely on."""
    src = os.path.join(CSRC, "probes", "mfma_f64_hazard.hip")
    if not force and os.path.exists(PROBE_PATH) and os.path.getmtime(PROBE_PATH) >= os.path.getmtime(src):
        return PROBE_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                           "-o", PROBE_PATH, src])
    return PROBE_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    build_probes(force="--force" in sys.argv)
    print(LIB_PATH)
