"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/pympc_hip.h declares; no
compute call is made (no GPU here)."""
import os
import re

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
    assert lib.pympc_nnls_workspace_doubles(64, 2048, 3) > 0 and lib.pympc_nnls_workspace_doubles(65, 10, 3) == 0
    assert lib.pympc_pnnls_workspace_doubles(64, 100, 64, 2) > 0 and lib.pympc_pnnls_workspace_doubles(64, 100, 65, 2) == 0


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
