"""Host-to-host feedback pipeline (pympc_amd/pipeline.py): uploads, solves and downloads of neighbouring batches overlap on
three streams; results must be those of the plain batched call (which the other GPU tests pin to the oracle and to the
reference's golden vectors), whatever the order the transfers complete in."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def test_pipeline_results_equal_direct_solve():
    from pympc_amd.optimization.solvers.hip import PreparedMpqp
    from pympc_amd.pipeline import FeedbackPipeline, pick_copy_streams
    m = golden("c2_model")
    prep = PreparedMpqp(m["Huu"], m["Hux"], m["Hxx"], m["fu"], m["fx"], float(m["g"]), m["Au"], m["Ax"], m["b"])
    B = 50000
    rng = np.random.RandomState(2)
    batches = [torch.as_tensor(rng.uniform(-s * m["x_max"], s * m["x_max"], (B, 4))).pin_memory() for s in (0.3, 1.0, 0.5, 0.3, 0.8)]
    pipe = FeedbackPipeline(prep, B, nu=1, depth=2)
    assert set(pipe.stream_report) >= {"candidates", "independent_of_compute", "independent_pair"}
    got = []
    slots = []
    for t, Xh in enumerate(batches):
        slots.append(pipe.submit(Xh))
        if t >= 1:                       # results of the previous batch while this one is in flight
            u, st = pipe.wait(slots[t - 1])
            got.append((u.clone(), st.clone()))
    u, st = pipe.wait(slots[-1])
    got.append((u.clone(), st.clone()))
    torch.cuda.synchronize()
    for Xh, (u, st) in zip(batches, got):
        # (the same outputs as the pipeline asks for: bitwise equality needs the same launch, see include/pympc_hip.h on
        # the active-set words as hand-over space between the kernels)
        ref = prep.solve_batch(Xh.cuda(), n_out=1, want_mult=False, want_active=True)
        assert np.array_equal(st.numpy(), ref.numpy("status"))
        assert np.array_equal(u.numpy(), ref.numpy("argmin"), equal_nan=True)
    # the stream probe tells streams that share a hardware queue from streams that do not: a stream never runs ahead of itself
    from pympc_amd.pipeline import _independent
    s = torch.cuda.Stream()
    assert not _independent(s, s)
    a, b, rep = pick_copy_streams()
    assert a is not b
