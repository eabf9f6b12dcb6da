"""Solver dispatch, mirroring /root/reference/pympc/optimization/programs.py:9-105.

The reference switches on a string (`solver='pnnls' | 'gurobi' | 'drake'`, programs.py:48-55, 98-105); this module
is the same switch with ONE backend, `'hip'`, which is also the default -- none of the reference's callers pass
`solver=` (polyhedron.py:444,523,756, parametric_programs.py:173, ...), so the default string is the global
switch.  `'pnnls'` is accepted as an alias so that code written against the reference runs unchanged; there is
no CPU path behind it.  `'hip_kkt'` selects, for LPs, the reference's OWN formulation on the device (the KKT system
of pnnls.py:109-124 solved as a PNNLS and decided by its residual, :137; 1 + m + n <= 1024): the default LP kernel is a
dual active-set method that decides feasibility row by row, and the two differ for LPs that are infeasible or
unbounded by a margin between 1e-9 and 1e-7.
"""
from .solvers import hip as _hip_backend

_ALIASES = ('hip', 'pnnls')


def linear_program(f, A, b, C=None, d=None, solver='hip'):
    """min_x f'x s.t. A x <= b, C x = d (programs.py:9-55).  Returns the reference's sol dict."""
    if solver in _ALIASES:
        from .solvers import hip_lp
        return hip_lp.linear_program(f, A, b, C, d)
    if solver == 'hip_kkt':
        from .solvers import hip_lp
        return hip_lp.linear_program_pnnls(f, A, b, C, d)
    raise ValueError('unknown solver ' + str(solver) + '.')


def quadratic_program(H, f, A, b, C=None, d=None, solver='hip'):
    """min_x .5 x'Hx + f'x s.t. A x <= b, C x = d, H > 0 (programs.py:57-105).  Returns the reference's sol dict."""
    if solver in _ALIASES or solver == 'hip_kkt':
        return _hip_backend.quadratic_program(H, f, A, b, C, d)
    raise ValueError('unknown solver ' + str(solver) + '.')
