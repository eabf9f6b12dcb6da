"""Scalar numpy model of the arithmetic the HIP wave kernel performs (tests only).

The HIP kernel (pympc_amd/csrc/qp_wave.hpp) does NOT keep a transformed copy of the NNLS matrix per
problem the way Lawson-Hanson's Fortran does; it keeps, per problem,
    Qt  (R x R)  orthogonal, Qt A_P = [Rk; 0]
    Wi  (k x k)  = Rk^-1 (upper triangular), updated in place
    cv  (R)      = Qt b
and reads the untransformed shared matrix for the dual sweep w = A'(Qt' [0; cv[k:]]).
Adding a column is one Householder reflector on Qt plus one new column of Wi; dropping position p is a
sequence of Givens rotations chosen from row p of Wi (they zero it), applied to Wi's columns, Qt's rows
and cv.  This file restates exactly that so the update formulas can be checked on the CPU against the
oracle before/without a GPU.  Selection rules are Lawson-Hanson's (first maximum of w, 0.01 independence
factor, positive trial coefficient, alpha interpolation, itmax = 3*cols).
"""
import numpy as np


def nnls_invqr(A, b, itmax=None):
    A = np.asarray(A, float); b = np.asarray(b, float)
    R, m = A.shape
    itmax = 3 * m if itmax is None else itmax
    Qt = np.eye(R); Wi = np.zeros((R, R)); cv = b.copy()
    P = []; y = []          # positions
    k = 0; it = 0; outer = 0
    inP = np.zeros(m, bool)
    status = 0
    while k < R and k < m:
        r = Qt[k:].T.dot(cv[k:])
        w = A.T.dot(r)
        w[inP] = -1.0
        accepted = False
        while True:
            j = int(np.argmax(w))
            if not (w[j] > 0.0):
                break
            q = Qt.dot(A[:, j])
            unorm = np.sqrt(np.sum(q[:k] ** 2))
            sigma = np.sqrt(np.sum(q[k:] ** 2))
            ok = False
            if sigma > 0.0:
                diag = -sigma if q[k] > 0 else sigma
                up = q[k] - diag
                if (unorm + abs(diag) * 0.01) - unorm > 0.0:
                    bb = up * diag
                    sm = (cv[k] * up + np.dot(cv[k + 1:], q[k + 1:])) / bb
                    ck = cv[k] + sm * up
                    if ck / diag > 0.0:
                        ok = True
            if ok:
                accepted = True
                break
            w[j] = 0.0
        if not accepted:
            break
        outer += 1
        v = q.copy(); v[:k] = 0.0; v[k] = up
        cv[k + 1:] += sm * q[k + 1:]; cv[k] = ck
        s = v.dot(Qt) / bb
        Qt += np.outer(v, s)
        newcol = -(np.triu(Wi[:k, :k]).dot(q[:k])) / diag
        Wi[:k, k] = newcol; Wi[k, :k] = 0.0; Wi[k, k] = 1.0 / diag
        P.append(j); y.append(0.0); inP[j] = True; k += 1
        z = np.triu(Wi[:k, :k]).dot(cv[:k])
        while True:
            it += 1
            if it > itmax:
                status = 3
                break
            yv = np.array(y)
            bad = z <= 0.0
            if not bad.any():
                y = list(z)
                break
            t = np.where(bad, yv / np.where(bad, yv - z, 1.0), np.inf)
            jj = int(np.argmin(t)); alpha = t[jj]
            yv = yv + alpha * (z - yv)
            y = list(yv)
            while True:
                # drop position jj
                for l in range(jj, k - 1):
                    xx, yy = Wi[jj, l], Wi[jj, l + 1]
                    rho = np.hypot(xx, yy)
                    cc, ss = (yy / rho, xx / rho) if rho > 0 else (1.0, 0.0)
                    cl = Wi[:k, l].copy(); cl1 = Wi[:k, l + 1].copy()
                    Wi[:k, l] = cc * cl - ss * cl1; Wi[:k, l + 1] = ss * cl + cc * cl1
                    rl = Qt[l].copy(); rl1 = Qt[l + 1].copy()
                    Qt[l] = cc * rl - ss * rl1; Qt[l + 1] = ss * rl + cc * rl1
                    a0, a1 = cv[l], cv[l + 1]
                    cv[l] = cc * a0 - ss * a1; cv[l + 1] = ss * a0 + cc * a1
                Wi[jj:k - 1, :k - 1] = Wi[jj + 1:k, :k - 1]
                inP[P[jj]] = False
                del P[jj]; del y[jj]; k -= 1
                nxt = [l for l in range(k) if y[l] <= 0.0]
                if not nxt:
                    break
                jj = nxt[0]
            z = np.triu(Wi[:k, :k]).dot(cv[:k])
        if status:
            break
    x = np.zeros(m)
    for l in range(k):
        x[P[l]] = y[l]
    rnorm = np.sqrt(np.sum(cv[k:] ** 2))
    return x, rnorm, (outer, it), status


def qp_via_nnls(H, f, A, b, tol=1e-7):
    """pnnls.py:149-241 with the model NNLS (inequalities only)."""
    L = np.linalg.cholesky(H); Li = np.linalg.inv(L); Hi = Li.T.dot(Li)
    M = A.dot(Li.T); mvec = b + A.dot(Hi).dot(f)
    An = np.vstack((-M.T, -mvec)); bn = np.concatenate((np.zeros(H.shape[0]), [1.0]))
    yv, r, iters, status = nnls_invqr(An, bn)
    if status:
        return dict(status=3)
    if r > tol:
        lam = yv / (1.0 + mvec.dot(yv))
        x = -Hi.dot(f + A.T.dot(lam))
        return dict(status=0, argmin=x, min=.5 * x.dot(H).dot(x) + f.dot(x), mult=lam, active=lam > tol, iters=iters)
    return dict(status=1, iters=iters)


# =====================================================================================================
# LP: numpy model of the HIP LP kernel (pympc_amd/csrc/lp_block.hpp)
# =====================================================================================================
def lp_dual_active_set(f, A, b, tol=1e-7, maxit=None):
    """min f'x s.t. A x <= b  (inequalities only; equalities folded as +-rows by the caller).

    The reference solves the LP's KKT system as one big partially-non-negative least squares problem
    (pnnls.py:41-147, (1+m+n) x 2m dense).  Its minimiser, when the residual is zero, is a primal-dual optimal pair
    (x, y); when the LP is infeasible or unbounded the residual is positive and every output is None.  The kernel
    computes the same pair with an active-set method that only ever factorises n x k matrices (k <= n):

      phase 1  Lawson-Hanson NNLS on the dual-feasibility system  min || A'y + f ||, y >= 0  (n rows, m columns);
               residual > tol  <=>  no dual feasible point  <=>  LP unbounded or infeasible  -> None
      phase 2  dual-simplex pivots on the same factorisation Qt A_W' = [R;0], Wi = R^-1:
               x = Qt'[Wi' b_W; 0] (minimum-norm point of the active face), most violated row e enters;
               independent of W -> joins with y_e = 0; dependent -> ratio test on d = Wi (Qt a_e)[:k], the blocking
               row leaves; no blocking row -> primal infeasible -> None.
    Returns dict(status, argmin, min, mult, active, iters).
    """
    f = np.asarray(f, float); A = np.asarray(A, float); b = np.asarray(b, float)
    m, n = A.shape
    maxit = 10 * (m + n) + 100 if maxit is None else maxit
    # ---- phase 1
    y, r, iters, st = nnls_invqr(A.T, -f)
    if st:
        return dict(status=3)
    scale_f = max(1.0, np.linalg.norm(f))
    if not (r < tol):
        return dict(status=1, phase=1, iters=iters)
    W = [int(j) for j in np.where(y > 0)[0]]
    yW = [float(y[j]) for j in W]

    def factor(Wl):
        k = len(Wl)
        if k == 0:
            return np.eye(n), np.zeros((0, 0))
        Q, Rm = np.linalg.qr(A[Wl].T, mode='complete')
        return Q.T, Rm[:k, :k]

    it = 0
    anorm = np.linalg.norm(A, axis=1)
    while True:
        it += 1
        if it > maxit:
            return dict(status=3)
        k = len(W)
        Qt, Rm = factor(W)
        if k:
            x = Qt[:k].T.dot(np.linalg.solve(Rm.T, b[W]))
        else:
            x = np.zeros(n)
        s = b - A.dot(x)
        s_rel = s / np.maximum(1.0, np.abs(b))
        s_rel[W] = np.inf
        e = int(np.argmin(s_rel))
        if s_rel[e] >= -1e-9:
            break
        q = Qt.dot(A[e])
        if np.linalg.norm(q[k:]) > 1e-9 * max(anorm[e], 1e-300) and k < n:
            W.append(e); yW.append(0.0)
            continue
        d = np.linalg.solve(Rm, q[:k]) if k else np.zeros(0)
        ratios = [(yW[i] / d[i], i) for i in range(k) if d[i] > 1e-12 * max(1.0, np.abs(d).max())]
        if not ratios:
            return dict(status=1, phase=2, iters=(iters, it))
        t, l = min(ratios)
        for i in range(k):
            yW[i] = max(yW[i] - t * d[i], 0.0)
        del W[l]; del yW[l]
        W.append(e); yW.append(t)
    mult = np.zeros(m)
    for i, j in enumerate(W):
        mult[j] = yW[i]
    return dict(status=0, argmin=x, min=float(f.dot(x)), mult=mult, active=mult > tol, iters=(iters, it), W=W)
