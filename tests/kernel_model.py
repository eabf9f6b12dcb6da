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
class InvQR(object):
    """The factorisation both kernels keep per problem:  Qt (R x R orthogonal), Qt G_W = [Rk; 0] for the k working
    columns G_W of a matrix with R rows, Wi = Rk^-1, cv = Qt c.  add() / drop() are the two update primitives."""

    def __init__(self, R, c):
        self.R = R
        self.Qt = np.eye(R); self.Wi = np.zeros((R, R)); self.cv = np.array(c, float)
        self.k = 0

    def trial(self, g):
        """Householder data for appending column g: returns None when g is (numerically) dependent on the working
        columns (Lawson-Hanson's 0.01 test), else (q, diag, up, bb, sm, ck) with ck / diag the trial coefficient."""
        k = self.k
        q = self.Qt.dot(g)
        unorm = np.sqrt(np.sum(q[:k] ** 2)); sigma = np.sqrt(np.sum(q[k:] ** 2))
        if not sigma > 0.0:
            return None
        diag = -sigma if q[k] > 0 else sigma
        up = q[k] - diag
        if not (unorm + abs(diag) * 0.01) - unorm > 0.0:
            return None
        bb = up * diag
        sm = (self.cv[k] * up + np.dot(self.cv[k + 1:], q[k + 1:])) / bb
        ck = self.cv[k] + sm * up
        return q, diag, up, bb, sm, ck

    def add(self, tr):
        q, diag, up, bb, sm, ck = tr
        k = self.k
        v = q.copy(); v[:k] = 0.0; v[k] = up
        self.cv[k + 1:] += sm * q[k + 1:]; self.cv[k] = ck
        s = v.dot(self.Qt) / bb
        self.Qt += np.outer(v, s)
        self.Wi[:k, k] = -(np.triu(self.Wi[:k, :k]).dot(q[:k])) / diag
        self.Wi[k, :k] = 0.0; self.Wi[k, k] = 1.0 / diag
        self.k += 1

    def drop(self, jj):
        Wi, Qt, cv, k = self.Wi, self.Qt, self.cv, self.k
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
        self.k -= 1

    def coeffs(self):
        k = self.k
        return np.triu(self.Wi[:k, :k]).dot(self.cv[:k])


def lp_invqr(f, A, b, tol=1e-7, maxit=None):
    """min f'x s.t. A x <= b  (inequalities only; equalities folded as +-rows by the caller).

    The reference solves the LP's KKT system as one big partially-non-negative least squares problem
    (pnnls.py:41-147, (1+m+n) x 2m dense).  Its minimiser, when the residual is zero, is a primal-dual optimal pair
    (x, y); when the LP is infeasible or unbounded the residual is positive and every output is None.  The kernel
    computes the same pair with an active-set method on the n x k factorisation above (k <= n working rows W):

      phase 1  Lawson-Hanson NNLS on the dual-feasibility system  min || A'y + f ||, y >= 0  (n rows, m columns);
               residual >= tol  <=>  no dual feasible point  <=>  LP unbounded or infeasible  -> None
      phase 2  dual active-set pivots: x = Qt'[Wi' b_W; 0] (minimum-norm point of {A_W x = b_W}); the most violated
               row e enters: independent of W -> appended (its multiplier starts at 0); dependent -> ratio test on
               d = Wi (Qt a_e)[:k], the blocking row leaves, e is appended; no blocking row -> primal infeasible.
               The multipliers are always y_W = Wi cv[:k], cv = Qt(-f) carried through both updates.
    Returns dict(status, argmin, min, mult, active, iters).
    """
    f = np.asarray(f, float); A = np.asarray(A, float); b = np.asarray(b, float)
    m, n = A.shape
    itmax = 3 * m
    maxit = 10 * (m + n) + 100 if maxit is None else maxit
    fnorm = max(1.0, np.linalg.norm(f))
    F = InvQR(n, -f)
    W = []; y = []
    inW = np.zeros(m, bool)
    it = 0; outer = 0
    # ---- phase 1 (same selection rules as nnls_invqr)
    while F.k < n and F.k < m:
        if np.sqrt(np.sum(F.cv[F.k:] ** 2)) <= 1e-13 * fnorm:
            break
        r = F.Qt[F.k:].T.dot(F.cv[F.k:])
        w = A.dot(r)
        w[inW] = -1.0
        tr = None
        while True:
            j = int(np.argmax(w))
            if not (w[j] > 0.0):
                break
            tr = F.trial(A[j])
            if tr is not None and tr[5] / tr[1] > 0.0:
                break
            tr = None
            w[j] = 0.0
        if tr is None:
            break
        outer += 1
        F.add(tr); W.append(j); y.append(0.0); inW[j] = True
        z = F.coeffs()
        while True:
            it += 1
            if it > itmax:
                return dict(status=3)
            yv = np.array(y)
            bad = z <= 0.0
            if not bad.any():
                y = list(z)
                break
            t = np.where(bad, yv / np.where(bad, yv - z, 1.0), np.inf)
            jj = int(np.argmin(t)); alpha = t[jj]
            y = list(yv + alpha * (z - yv))
            while True:
                F.drop(jj); inW[W[jj]] = False
                del W[jj]; del y[jj]
                nxt = [l for l in range(F.k) if y[l] <= 0.0]
                if not nxt:
                    break
                jj = nxt[0]
            z = F.coeffs()
    rnorm = np.sqrt(np.sum(F.cv[F.k:] ** 2))
    if not (rnorm < tol):
        return dict(status=1, phase=1, iters=(outer, it, 0))
    # ---- phase 2
    anorm = np.linalg.norm(A, axis=1)
    ainv = np.where(anorm > 0, 1.0 / np.where(anorm > 0, anorm, 1.0), 0.0)
    piv = 0
    while True:
        k = F.k
        x = F.Qt[:k].T.dot(np.triu(F.Wi[:k, :k]).T.dot(b[W])) if k else np.zeros(n)
        viol = (A.dot(x) - b)
        thr = 1e-9 * np.maximum(1.0, np.abs(b))
        score = np.where(viol > thr, (viol - thr) * np.where(ainv > 0, ainv, 1.0), -1.0)
        score[inW] = -1.0
        e = int(np.argmax(score))
        if not (score[e] > 0.0):
            break
        piv += 1
        if piv > maxit:
            return dict(status=3)
        tr = F.trial(A[e]) if k < n else None
        if tr is not None:   # stricter independence test than Lawson-Hanson's for the entering row
            un = np.sqrt(np.sum(tr[0][:k] ** 2))
            if not abs(tr[1]) > 1e-10 * (un + abs(tr[1])):
                tr = None
        if tr is None:
            q = F.Qt.dot(A[e])
            d = np.triu(F.Wi[:k, :k]).dot(q[:k]) if k else np.zeros(0)
            yv = F.coeffs()
            dmax = np.abs(d).max() if k else 0.0
            cand = [(max(yv[i], 0.0) / d[i], i) for i in range(k) if d[i] > 1e-12 * max(1.0, dmax)]
            if not cand:
                return dict(status=1, phase=2, iters=(outer, it, piv))
            _, l = min(cand)
            F.drop(l); inW[W[l]] = False; del W[l]
            tr = F.trial(A[e])
            if tr is None:
                return dict(status=3)
        F.add(tr); W.append(e); inW[e] = True
    yv = F.coeffs() if F.k else np.zeros(0)
    mult = np.zeros(m)
    for i, j in enumerate(W):
        mult[j] = max(yv[i], 0.0)
    return dict(status=0, argmin=x, min=float(f.dot(x)), mult=mult, active=mult > tol, iters=(outer, it, piv), W=list(W))
