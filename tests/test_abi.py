"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/pympc_hip.h declares; no
compute call is made (no GPU here)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "pympc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pympc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from pympc_amd import _hip
    assert _declared() == sorted(_hip.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from pympc_amd import _hip
    from pympc_amd.build import build
    build()
    lib = _hip.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.pympc_hip_abi_version() >= 8   # 8: pympc_point_location with decided cells and decision lists
    nmax, mmax = _hip.limits()
    assert nmax >= 20 and mmax >= 370
    assert lib.pympc_mpqp_workspace_doubles(20, 370, 4) > 20 * 370 + 384 * 384   # 370 rows padded to 384 columns
    assert lib.pympc_mpqp_workspace_doubles(1000, 370, 4) == 0
    # shapes inside the variable / row limits whose parameter count does not fit: refused, never a raw launch error
    assert lib.pympc_mpqp_workspace_doubles(20, 370, 65) == 0          # the parameter vector is staged one entry per lane
    assert lib.pympc_mpqp_workspace_doubles(63, 100, 4) > 0
    assert lib.pympc_mpqp_workspace_doubles(63, 100, 8) == 0           # factor matrices + Hux beyond 160 KiB of LDS
    assert lib.pympc_mpqp_multi_workspace_doubles(63, 100, 8, 10) == 0
    assert lib.pympc_nnls_workspace_doubles(64, 2048, 3) > 0 and lib.pympc_nnls_workspace_doubles(1025, 10, 3) == 0
    assert lib.pympc_nnls_workspace_doubles(65, 10, 3) >= 65 * 10 * 3             # 65 .. 1024 rows: a working copy per problem
    assert lib.pympc_pnnls_workspace_doubles(64, 100, 64, 2) > 0 and lib.pympc_pnnls_workspace_doubles(64, 100, 257, 2) == 0
    assert lib.pympc_pnnls_workspace_doubles(64, 100, 65, 2) > lib.pympc_pnnls_workspace_doubles(64, 100, 64, 2)   # 65 .. 256 free variables: workgroup kernels


def test_no_gpu_means_loud_failure():
    import torch
    from pympc_amd import _hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_hip.HipLibraryError):
        _hip.device()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pympc_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pympc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("# the cpu oracle", ""), os.path.join(dirpath, f)


def test_decision_list_assembly_matches_a_plain_loop():
    """Host logic of the point-location table (ExplicitSolution._assemble_lists, numpy only): the vectorised assembly of the
    per-cell decision lists against a straightforward loop, on random candidates -- truncation at the first candidate
    without a crossing row, the length cap, offsets and codes."""
    from pympc_amd.optimization.parametric_programs import ExplicitSolution
    rng = np.random.RandomState(0)
    cap = 48
    ncell = 400
    cu, cr, ck, ri = [], [], [], []
    for u in range(ncell):
        if u % 17 == 5:
            continue                                      # a cell without candidates
        for r in np.sort(rng.choice(50, rng.randint(1, 7), replace=False)):
            k = rng.randint(0, 12) if rng.rand() < .8 else 0
            cu.append(u); cr.append(int(r)); ck.append(k)
            ri += list(np.sort(rng.choice(1000, k, replace=False)))
    cu, cr, ck, ri = (np.array(v, dtype=np.int64) for v in (cu, cr, ck, ri))
    pool, codes = ExplicitSolution._assemble_lists(cu, cr, ck, ri, ncell, cap)
    start = np.searchsorted(cu, np.arange(ncell + 1))
    rstart = np.concatenate(([0], np.cumsum(ck)))
    ref_pool, ref_codes = [0], np.full(ncell, -2, np.int64)
    for u in range(ncell):
        ints = [0]
        for j in range(start[u], start[u + 1]):
            ints += [int(cr[j]), int(ck[j])] + ri[rstart[j]:rstart[j + 1]].tolist()
            ints[0] += 1
            if ck[j] == 0:
                break
        if start[u + 1] > start[u] and len(ints) <= cap:
            ref_codes[u] = -3 - len(ref_pool)
            ref_pool += ints
    assert np.array_equal(pool, np.array(ref_pool, dtype=np.int32))
    assert np.array_equal(codes, ref_codes)
    assert (codes == -2).sum() > 10 and (codes <= -3).sum() > 200
