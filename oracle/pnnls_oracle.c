/*
 * oracle/pnnls_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement of pympc's pnnls hot path, used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under pympc_amd/ may import, link or call it.
 *
 * What is restated (reference file:line under /root/reference):
 *   oracle_qp        pympc/optimization/solvers/pnnls.py:149-241  (QP -> NNLS, Bemporad 2016 Thm 1)
 *   oracle_pnnls     pympc/optimization/solvers/pnnls.py:5-39     (PNNLS via pinv projection)
 *   oracle_lp        pympc/optimization/solvers/pnnls.py:41-147   (LP -> KKT -> PNNLS)
 *   oracle_mpqp_solve_batch
 *                    pympc/optimization/parametric_programs.py:155-179 looped as
 *                    pympc/control/controllers.py:76-124 does (feedforward / feedback)
 *   oracle_point_location
 *                    pympc/optimization/parametric_programs.py:442-532, 349-400 and
 *                    pympc/geometry/polyhedron.py:578-604 (contains)
 *   oracle_condense  pympc/dynamics/discrete_time_systems.py:676-751 (condense_pwa_system) and
 *                    pympc/control/controllers.py:682-749 (condense_optimal_control_problem)
 *   oracle_nnls      scipy.optimize.nnls -- THIRD-PARTY, absent from /root/reference.
 *                    The reference pins scipy<=1.11.4 (setup.py:18, requirements.txt:1), i.e. the
 *                    f2py-wrapped Fortran NNLS of Lawson & Hanson, "Solving Least Squares Problems"
 *                    (1974), ch. 23.  This file restates that published algorithm: Householder
 *                    add-column, Givens delete-column, 0.01 independence factor, first-maximum
 *                    dual selection, itmax = 3*cols.  numpy.linalg.pinv (rcond 1e-15, SVD) is
 *                    restated with a one-sided Jacobi SVD; numpy.linalg.cholesky / inv with a
 *                    plain Cholesky and a triangular inverse.
 *
 * Pinning: checked against the reference's own known-answer tests
 * (pympc/test/test_optimization/test_solvers.py:12-345 etc., transcribed as data in tests/) and
 * against KKT-certified outputs of the reference run in the build container
 * (tests/golden/, written by tests/golden/make_golden.py).
 *
 * All matrices are row-major (numpy C order), fp64.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_INFEASIBLE 1 /* QP infeasible / LP infeasible-or-unbounded: reference returns all-None */
#define ORACLE_MAXITER 3    /* scipy raises RuntimeError */
#define ORACLE_NOT_PD -2    /* numpy.linalg.LinAlgError from cholesky */

/* ------------------------------------------------------------------------------------------- */
/* Householder (construct / apply) and Givens pieces of Lawson-Hanson                           */
/* ------------------------------------------------------------------------------------------- */

/* Build the reflector that maps u[p], u[l1..m-1] onto a multiple of e_p. u is overwritten at p. */
static void house_make(double *u, int p, int l1, int m, double *up)
{
    double cl = fabs(u[p]);
    for (int j = l1; j < m; ++j)
        if (fabs(u[j]) > cl) cl = fabs(u[j]);
    if (cl <= 0.0) { *up = 0.0; return; }
    double clinv = 1.0 / cl;
    double sm = (u[p] * clinv) * (u[p] * clinv);
    for (int j = l1; j < m; ++j) sm += (u[j] * clinv) * (u[j] * clinv);
    cl *= sqrt(sm);
    if (u[p] > 0.0) cl = -cl;
    *up = u[p] - cl;
    u[p] = cl;
}

/* Apply the reflector (u, up) built at pivot p to the vector c. */
static void house_apply(const double *u, double up, int p, int l1, int m, double *c)
{
    double b = up * u[p];
    if (b >= 0.0) return;
    b = 1.0 / b;
    double sm = c[p] * up;
    for (int i = l1; i < m; ++i) sm += c[i] * u[i];
    if (sm != 0.0) {
        sm *= b;
        c[p] += sm * up;
        for (int i = l1; i < m; ++i) c[i] += sm * u[i];
    }
}

static void givens_make(double a, double b, double *c, double *s, double *sig)
{
    if (fabs(a) > fabs(b)) {
        double xr = b / a, yr = sqrt(1.0 + xr * xr);
        *c = copysign(1.0 / yr, a);
        *s = (*c) * xr;
        *sig = fabs(a) * yr;
    } else if (b != 0.0) {
        double xr = a / b, yr = sqrt(1.0 + xr * xr);
        *s = copysign(1.0 / yr, b);
        *c = (*s) * xr;
        *sig = fabs(b) * yr;
    } else {
        *sig = 0.0; *c = 0.0; *s = 1.0;
    }
}

/*
 * Lawson-Hanson NNLS:  min ||A x - b||_2  s.t. x >= 0.   A is rows x cols, row-major.
 * x (cols) and *rnorm are written; iters[0] = outer (add-column) iterations, iters[1] = inner
 * (secondary-loop) passes.  Returns ORACLE_OK or ORACLE_MAXITER.
 */
int oracle_nnls(const double *A_rm, int rows, int cols, const double *b_in, double *x, double *rnorm, int *iters)
{
    const int m = rows, n = cols;
    double *a = (double *)malloc(sizeof(double) * (size_t)m * (size_t)(n > 0 ? n : 1)); /* column-major work copy */
    double *b = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double *w = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    double *zz = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    int *index = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; ++j)
        for (int l = 0; l < m; ++l) a[(size_t)j * m + l] = A_rm[(size_t)l * n + j];
    for (int l = 0; l < m; ++l) b[l] = b_in[l];
    for (int j = 0; j < n; ++j) { x[j] = 0.0; index[j] = j; }
#define COL(j) (a + (size_t)(j) * m)
    const int itmax = 3 * n;
    int iter = 0, outer = 0, mode = ORACLE_OK;
    int iz1 = 0, iz2 = n - 1, nsetp = 0; /* P = index[0..nsetp-1], Z = index[iz1..iz2]; npp1 == nsetp (0-based) */
    const double factor = 0.01;

    for (;;) {
        if (iz1 > iz2 || nsetp >= m) break;
        /* dual vector on Z */
        for (int iz = iz1; iz <= iz2; ++iz) {
            int j = index[iz];
            double sm = 0.0;
            for (int l = nsetp; l < m; ++l) sm += COL(j)[l] * b[l];
            w[j] = sm;
        }
        int j = -1, izmax = -1;
        double up = 0.0;
        int accepted = 0;
        for (;;) {
            double wmax = 0.0;
            izmax = -1;
            for (int iz = iz1; iz <= iz2; ++iz) {
                int jj = index[iz];
                if (w[jj] > wmax) { wmax = w[jj]; izmax = iz; }
            }
            if (wmax <= 0.0) break; /* Kuhn-Tucker satisfied */
            j = index[izmax];
            double asave = COL(j)[nsetp];
            house_make(COL(j), nsetp, nsetp + 1, m, &up);
            double unorm = 0.0;
            for (int l = 0; l < nsetp; ++l) unorm += COL(j)[l] * COL(j)[l];
            unorm = sqrt(unorm);
            volatile double t1 = unorm + fabs(COL(j)[nsetp]) * factor;
            if (t1 - unorm > 0.0) {
                for (int l = 0; l < m; ++l) zz[l] = b[l];
                house_apply(COL(j), up, nsetp, nsetp + 1, m, zz);
                double ztest = zz[nsetp] / COL(j)[nsetp];
                if (ztest > 0.0) { accepted = 1; break; }
            }
            COL(j)[nsetp] = asave;
            w[j] = 0.0;
        }
        if (!accepted) break;
        ++outer;
        /* move j from Z to P */
        for (int l = 0; l < m; ++l) b[l] = zz[l];
        index[izmax] = index[iz1];
        index[iz1] = j;
        ++iz1;
        const int piv = nsetp;
        ++nsetp;
        for (int jz = iz1; jz <= iz2; ++jz) house_apply(COL(j), up, piv, nsetp, m, COL(index[jz]));
        for (int l = nsetp; l < m; ++l) COL(j)[l] = 0.0;
        w[j] = 0.0;
        /* solve the triangular system into zz */
#define TRISOLVE()                                                                  \
        do {                                                                        \
            int jj_ = -1;                                                           \
            for (int l_ = 0; l_ < nsetp; ++l_) {                                    \
                int ip_ = nsetp - 1 - l_;                                           \
                if (l_ != 0)                                                        \
                    for (int ii_ = 0; ii_ <= ip_; ++ii_) zz[ii_] -= COL(jj_)[ii_] * zz[ip_ + 1]; \
                jj_ = index[ip_];                                                   \
                zz[ip_] /= COL(jj_)[ip_];                                           \
            }                                                                       \
        } while (0)
        TRISOLVE();
        /* secondary loop */
        for (;;) {
            ++iter;
            if (iter > itmax) { mode = ORACLE_MAXITER; goto done; }
            double alpha = 2.0;
            int jj = -1;
            for (int ip = 0; ip < nsetp; ++ip) {
                int l = index[ip];
                if (zz[ip] <= 0.0) {
                    double t = -x[l] / (zz[ip] - x[l]);
                    if (alpha > t) { alpha = t; jj = ip; }
                }
            }
            if (alpha == 2.0) break;
            for (int ip = 0; ip < nsetp; ++ip) {
                int l = index[ip];
                x[l] += alpha * (zz[ip] - x[l]);
            }
            int i = index[jj];
            for (;;) {
                x[i] = 0.0;
                if (jj != nsetp - 1) {
                    ++jj;
                    for (int jc = jj; jc < nsetp; ++jc) {
                        int ii = index[jc];
                        index[jc - 1] = ii;
                        double cc, ss, sig;
                        givens_make(COL(ii)[jc - 1], COL(ii)[jc], &cc, &ss, &sig);
                        COL(ii)[jc - 1] = sig;
                        COL(ii)[jc] = 0.0;
                        for (int l = 0; l < n; ++l) {
                            if (l != ii) {
                                double t = COL(l)[jc - 1];
                                COL(l)[jc - 1] = cc * t + ss * COL(l)[jc];
                                COL(l)[jc] = -ss * t + cc * COL(l)[jc];
                            }
                        }
                        double t = b[jc - 1];
                        b[jc - 1] = cc * t + ss * b[jc];
                        b[jc] = -ss * t + cc * b[jc];
                    }
                }
                --nsetp;
                --iz1;
                index[iz1] = i;
                /* round-off guard: anything left in P that is non-positive leaves too */
                int again = 0;
                for (jj = 0; jj < nsetp; ++jj) {
                    i = index[jj];
                    if (x[i] <= 0.0) { again = 1; break; }
                }
                if (!again) break;
            }
            for (int l = 0; l < m; ++l) zz[l] = b[l];
            TRISOLVE();
        }
        for (int ip = 0; ip < nsetp; ++ip) x[index[ip]] = zz[ip];
    }
done:;
    double sm = 0.0;
    for (int l = nsetp; l < m; ++l) sm += b[l] * b[l];
    *rnorm = sqrt(sm);
    if (iters) { iters[0] = outer; iters[1] = iter; }
#undef COL
#undef TRISOLVE
    free(a); free(b); free(w); free(zz); free(index);
    return mode;
}

/* ------------------------------------------------------------------------------------------- */
/* small dense helpers (numpy.linalg restated)                                                  */
/* ------------------------------------------------------------------------------------------- */

/* lower Cholesky H = L L', row-major n x n; returns 0 or ORACLE_NOT_PD */
static int chol_lower(const double *H, int n, double *L)
{
    memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j) {
        double d = H[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.0)) return ORACLE_NOT_PD;
        d = sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = H[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / d;
        }
    }
    return 0;
}

/* inverse of a lower-triangular matrix */
static void tri_lower_inverse(const double *L, int n, double *Li)
{
    memset(Li, 0, sizeof(double) * (size_t)n * n);
    for (int c = 0; c < n; ++c) {
        Li[c * n + c] = 1.0 / L[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            double s = 0.0;
            for (int k = c; k < r; ++k) s += L[r * n + k] * Li[k * n + c];
            Li[r * n + c] = -s / L[r * n + r];
        }
    }
}

/*
 * Pseudo-inverse of B (rows x cols, rows >= 1, any rank) like numpy.linalg.pinv (rcond = 1e-15):
 * one-sided Jacobi SVD of B (Hestenes), singular values below rcond * s_max dropped.
 * Bp is cols x rows.
 */
static void pinv_jacobi(const double *B, int rows, int cols, double *Bp)
{
    double *U = (double *)malloc(sizeof(double) * (size_t)rows * cols); /* column-major copy, cols columns */
    double *V = (double *)calloc((size_t)cols * cols, sizeof(double));  /* column-major */
    double *s = (double *)malloc(sizeof(double) * (size_t)(cols > 0 ? cols : 1));
    for (int j = 0; j < cols; ++j) {
        for (int i = 0; i < rows; ++i) U[(size_t)j * rows + i] = B[(size_t)i * cols + j];
        V[(size_t)j * cols + j] = 1.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < cols - 1; ++p)
            for (int q = p + 1; q < cols; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                double *up = U + (size_t)p * rows, *uq = U + (size_t)q * rows;
                for (int i = 0; i < rows; ++i) { alpha += up[i] * up[i]; beta += uq[i] * uq[i]; gamma += up[i] * uq[i]; }
                if (gamma == 0.0 || fabs(gamma) <= 1e-16 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < rows; ++i) { double a = up[i], b = uq[i]; up[i] = c * a - sn * b; uq[i] = sn * a + c * b; }
                double *vp = V + (size_t)p * cols, *vq = V + (size_t)q * cols;
                for (int i = 0; i < cols; ++i) { double a = vp[i], b = vq[i]; vp[i] = c * a - sn * b; vq[i] = sn * a + c * b; }
            }
        if (!rotated) break;
    }
    double smax = 0.0;
    for (int j = 0; j < cols; ++j) {
        double nn = 0.0;
        for (int i = 0; i < rows; ++i) nn += U[(size_t)j * rows + i] * U[(size_t)j * rows + i];
        s[j] = sqrt(nn);
        if (s[j] > smax) smax = s[j];
    }
    memset(Bp, 0, sizeof(double) * (size_t)rows * cols);
    for (int j = 0; j < cols; ++j) {
        if (!(s[j] > 1e-15 * smax)) continue;
        double inv2 = 1.0 / (s[j] * s[j]); /* V_j (1/s_j) (U_j/s_j)' */
        for (int r = 0; r < cols; ++r) {
            double v = V[(size_t)j * cols + r] * inv2;
            if (v == 0.0) continue;
            for (int i = 0; i < rows; ++i) Bp[(size_t)r * rows + i] += v * U[(size_t)j * rows + i];
        }
    }
    free(U); free(V); free(s);
}

/* ------------------------------------------------------------------------------------------- */
/* pnnls.py:149-241                                                                             */
/* ------------------------------------------------------------------------------------------- */
/*
 * Strictly convex QP  min .5 x'Hx + f'x  s.t. A x <= b, C x = d  through NNLS.
 * Outputs (when status == ORACLE_OK): xmin[1], argmin[n], mult_ineq[n_ineq], mult_eq[n_eq],
 * active[n_ineq] (0/1 flags of multiplier > tol).  C/d may be NULL with n_eq == 0.
 * iters[2] as oracle_nnls.  Returns ORACLE_OK / ORACLE_INFEASIBLE / ORACLE_MAXITER / ORACLE_NOT_PD.
 */
int oracle_qp(const double *H, const double *f, const double *A, const double *b, const double *C, const double *d,
              int n, int n_ineq, int n_eq, double tol, double *xmin, double *argmin, double *mult_ineq,
              double *mult_eq, int *active, double *rnorm_out, int *iters)
{
    const int mt = n_ineq + 2 * n_eq;
    double *AC = (double *)malloc(sizeof(double) * (size_t)(mt > 0 ? mt : 1) * n);
    double *bd = (double *)malloc(sizeof(double) * (size_t)(mt > 0 ? mt : 1));
    memcpy(AC, A, sizeof(double) * (size_t)n_ineq * n);
    memcpy(bd, b, sizeof(double) * (size_t)n_ineq);
    for (int i = 0; i < n_eq; ++i) { /* pnnls.py:202-207 */
        for (int j = 0; j < n; ++j) {
            AC[(size_t)(n_ineq + i) * n + j] = C[(size_t)i * n + j];
            AC[(size_t)(n_ineq + n_eq + i) * n + j] = -C[(size_t)i * n + j];
        }
        bd[n_ineq + i] = d[i];
        bd[n_ineq + n_eq + i] = -d[i];
    }
    double *L = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *Li = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *Hi = (double *)malloc(sizeof(double) * (size_t)n * n);
    int st = chol_lower(H, n, L); /* :210 */
    if (st != 0) { free(AC); free(bd); free(L); free(Li); free(Hi); return st; }
    tri_lower_inverse(L, n, Li); /* :211 */
    for (int i = 0; i < n; ++i)  /* :212  H_inv = L_inv' L_inv */
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Li[k * n + i] * Li[k * n + j];
            Hi[i * n + j] = s;
        }
    /* :213-214  M = AC L_inv',  m = bd + AC H_inv f ; A_nnls = [-M'; -m'] is (n+1) x mt */
    double *An = (double *)malloc(sizeof(double) * (size_t)(n + 1) * (mt > 0 ? mt : 1));
    double *mv = (double *)malloc(sizeof(double) * (size_t)(mt > 0 ? mt : 1));
    double *Hif = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += Hi[i * n + k] * f[k];
        Hif[i] = s;
    }
    for (int r = 0; r < mt; ++r) {
        for (int c = 0; c < n; ++c) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += AC[(size_t)r * n + k] * Li[c * n + k];
            An[(size_t)c * mt + r] = -s;
        }
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += AC[(size_t)r * n + k] * Hif[k];
        mv[r] = bd[r] + s;
        An[(size_t)n * mt + r] = -mv[r];
    }
    double *bn = (double *)calloc((size_t)n + 1, sizeof(double));
    bn[n] = 1.0; /* gamma, :215-217 */
    double *y = (double *)malloc(sizeof(double) * (size_t)(mt > 0 ? mt : 1));
    double r = 0.0;
    st = oracle_nnls(An, n + 1, mt, bn, y, &r, iters); /* :218 */
    if (rnorm_out) *rnorm_out = r;
    if (st == ORACLE_OK) {
        if (r > tol) { /* :230 */
            double my = 0.0;
            for (int i = 0; i < mt; ++i) my += mv[i] * y[i];
            double den = 1.0 + my;
            for (int i = 0; i < mt; ++i) y[i] = y[i] / den; /* lam, :231 */
            for (int i = 0; i < n_ineq; ++i) { mult_ineq[i] = y[i]; active[i] = y[i] > tol; }
            double *g = (double *)malloc(sizeof(double) * (size_t)n);
            for (int j = 0; j < n; ++j) {
                double s = f[j];
                for (int i = 0; i < mt; ++i) s += AC[(size_t)i * n + j] * y[i];
                g[j] = s;
            }
            for (int i = 0; i < n; ++i) { /* :233 */
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += Hi[i * n + k] * g[k];
                argmin[i] = -s;
            }
            double q = 0.0, lin = 0.0; /* :234 */
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += argmin[k] * H[k * n + i];
                q += s * argmin[i];
                lin += f[i] * argmin[i];
            }
            *xmin = .5 * q + lin;
            for (int i = 0; i < n_eq; ++i) mult_eq[i] = y[n_ineq + i] - y[n_ineq + n_eq + i]; /* :236-239 */
            free(g);
        } else
            st = ORACLE_INFEASIBLE;
    }
    free(AC); free(bd); free(L); free(Li); free(Hi); free(An); free(mv); free(Hif); free(bn); free(y);
    return st;
}

/* ------------------------------------------------------------------------------------------- */
/* pnnls.py:5-39                                                                                */
/* ------------------------------------------------------------------------------------------- */
/* min ||A v + B u - c||  s.t. v >= 0.  A: rows x nv, B: rows x nu, c: rows. */
int oracle_pnnls(const double *A, const double *B, const double *c, int rows, int nv, int nu, double *v, double *u,
                 double *r, int *iters)
{
    double *Bp = (double *)malloc(sizeof(double) * (size_t)rows * (nu > 0 ? nu : 1));
    pinv_jacobi(B, rows, nu, Bp); /* :30 */
    double *Bbar = (double *)malloc(sizeof(double) * (size_t)rows * rows);
    for (int i = 0; i < rows; ++i) /* :31 */
        for (int j = 0; j < rows; ++j) {
            double s = 0.0;
            for (int k = 0; k < nu; ++k) s += B[(size_t)i * nu + k] * Bp[(size_t)k * rows + j];
            Bbar[(size_t)i * rows + j] = (i == j ? 1.0 : 0.0) - s;
        }
    double *Abar = (double *)calloc((size_t)rows * (nv > 0 ? nv : 1), sizeof(double));
    double *bbar = (double *)calloc((size_t)rows, sizeof(double));
    for (int i = 0; i < rows; ++i) { /* :32-33 */
        for (int k = 0; k < rows; ++k) {
            double bik = Bbar[(size_t)i * rows + k];
            if (bik == 0.0) continue;
            const double *Ak = A + (size_t)k * nv;
            double *Ai = Abar + (size_t)i * nv;
            for (int j = 0; j < nv; ++j) Ai[j] += bik * Ak[j];
            bbar[i] += bik * c[k];
        }
    }
    int st = oracle_nnls(Abar, rows, nv, bbar, v, r, iters); /* :36 */
    double *t = (double *)malloc(sizeof(double) * (size_t)rows);
    for (int i = 0; i < rows; ++i) { /* :37 */
        double s = -c[i];
        for (int j = 0; j < nv; ++j) s += A[(size_t)i * nv + j] * v[j];
        t[i] = s;
    }
    for (int k = 0; k < nu; ++k) {
        double s = 0.0;
        for (int i = 0; i < rows; ++i) s += Bp[(size_t)k * rows + i] * t[i];
        u[k] = -s;
    }
    free(Bp); free(Bbar); free(Abar); free(bbar); free(t);
    return st;
}

/* ------------------------------------------------------------------------------------------- */
/* pnnls.py:41-147                                                                              */
/* ------------------------------------------------------------------------------------------- */
int oracle_lp(const double *f, const double *A, const double *b, const double *C, const double *d, int n, int n_ineq,
              int n_eq, double tol, double *xmin, double *argmin, double *mult_ineq, double *mult_eq, int *active,
              double *rnorm_out, int *iters)
{
    const int mt = n_ineq + 2 * n_eq;
    const int rows = 1 + mt + n;
    double *AC = (double *)malloc(sizeof(double) * (size_t)(mt > 0 ? mt : 1) * n);
    double *bd = (double *)malloc(sizeof(double) * (size_t)(mt > 0 ? mt : 1));
    memcpy(AC, A, sizeof(double) * (size_t)n_ineq * n);
    memcpy(bd, b, sizeof(double) * (size_t)n_ineq);
    for (int i = 0; i < n_eq; ++i) { /* :101-106 */
        for (int j = 0; j < n; ++j) {
            AC[(size_t)(n_ineq + i) * n + j] = C[(size_t)i * n + j];
            AC[(size_t)(n_ineq + n_eq + i) * n + j] = -C[(size_t)i * n + j];
        }
        bd[n_ineq + i] = d[i];
        bd[n_ineq + n_eq + i] = -d[i];
    }
    /* :109-124 */
    double *Ap = (double *)calloc((size_t)rows * 2 * (mt > 0 ? mt : 1), sizeof(double));
    double *Bp = (double *)calloc((size_t)rows * n, sizeof(double));
    double *cp = (double *)calloc((size_t)rows, sizeof(double));
    const int nv = 2 * mt;
    for (int i = 0; i < mt; ++i) Ap[i] = bd[i];                                  /* row 0: [bd, 0] */
    for (int i = 0; i < mt; ++i) Ap[(size_t)(1 + i) * nv + mt + i] = 1.0;        /* rows 1..mt: [0, I] */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < mt; ++i) Ap[(size_t)(1 + mt + j) * nv + i] = AC[(size_t)i * n + j]; /* [AC', 0] */
    for (int j = 0; j < n; ++j) Bp[j] = f[j];
    for (int i = 0; i < mt; ++i)
        for (int j = 0; j < n; ++j) Bp[(size_t)(1 + i) * n + j] = AC[(size_t)i * n + j];
    for (int i = 0; i < mt; ++i) cp[1 + i] = bd[i];
    for (int j = 0; j < n; ++j) cp[1 + mt + j] = -f[j];
    double *ys = (double *)malloc(sizeof(double) * (size_t)(nv > 0 ? nv : 1));
    double r = 0.0;
    int st = oracle_pnnls(Ap, Bp, cp, rows, nv, n, ys, argmin, &r, iters); /* :125 */
    if (rnorm_out) *rnorm_out = r;
    if (st == ORACLE_OK) {
        if (r < tol) { /* :137 */
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += f[j] * argmin[j];
            *xmin = s;
            for (int i = 0; i < n_ineq; ++i) { mult_ineq[i] = ys[i]; active[i] = ys[i] > tol; }
            for (int i = 0; i < n_eq; ++i) mult_eq[i] = ys[n_ineq + i] - ys[n_ineq + n_eq + i];
        } else
            st = ORACLE_INFEASIBLE;
    }
    free(AC); free(bd); free(Ap); free(Bp); free(cp); free(ys);
    return st;
}

/* ------------------------------------------------------------------------------------------- */
/* parametric_programs.py:155-179 in the loop of controllers.py:76-124                          */
/* ------------------------------------------------------------------------------------------- */
/*
 * For each state x_k (B states, p each):  f = Hux x + fu ; b = b0 - Ax x ; quadratic_program(Huu, f, Au, b);
 * min += .5 x'Hxx x + fx'x + g.  Exactly as the reference does, the Cholesky factor, its inverse and
 * M are recomputed for every state (pnnls.py:210-213) -- this is the "reference-as-is" CPU baseline.
 * status[k] as oracle_qp; argmin [B x n], vmin [B], mult [B x m] (may be NULL), active [B x m] (may be NULL).
 */
int oracle_mpqp_solve_batch(const double *Huu, const double *Hux, const double *Hxx, const double *fu, const double *fx,
                            double g, const double *Au, const double *Ax, const double *b0, int n, int m, int p,
                            const double *X, long B, double tol, double *argmin, double *vmin, double *mult,
                            int *active, int *status, int *iters)
{
    double *f = (double *)malloc(sizeof(double) * (size_t)n);
    double *b = (double *)malloc(sizeof(double) * (size_t)m);
    double *lam = (double *)malloc(sizeof(double) * (size_t)m);
    int *act = (int *)malloc(sizeof(int) * (size_t)m);
    for (long k = 0; k < B; ++k) {
        const double *x = X + (size_t)k * p;
        for (int i = 0; i < n; ++i) {
            double s = fu[i];
            for (int j = 0; j < p; ++j) s += Hux[(size_t)i * p + j] * x[j];
            f[i] = s;
        }
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int j = 0; j < p; ++j) s += Ax[(size_t)i * p + j] * x[j];
            b[i] = b0[i] - s;
        }
        double v = 0.0, rn = 0.0;
        int it[2] = {0, 0};
        int st = oracle_qp(Huu, f, Au, b, NULL, NULL, n, m, 0, tol, &v, argmin + (size_t)k * n, lam, NULL, act, &rn, it);
        status[k] = st;
        if (iters) { iters[2 * k] = it[0]; iters[2 * k + 1] = it[1]; }
        if (st == ORACLE_OK) {
            double q = 0.0, lin = 0.0;
            for (int i = 0; i < p; ++i) {
                double s = 0.0;
                for (int j = 0; j < p; ++j) s += x[j] * Hxx[(size_t)j * p + i];
                q += s * x[i];
                lin += fx[i] * x[i];
            }
            vmin[k] = v + .5 * q + lin + g;
            if (mult) memcpy(mult + (size_t)k * m, lam, sizeof(double) * (size_t)m);
            if (active) memcpy(active + (size_t)k * m, act, sizeof(int) * (size_t)m);
        } else {
            vmin[k] = NAN;
            for (int i = 0; i < n; ++i) argmin[(size_t)k * n + i] = NAN;
            if (mult) for (int i = 0; i < m; ++i) mult[(size_t)k * m + i] = NAN;
            if (active) memset(active + (size_t)k * m, 0, sizeof(int) * (size_t)m);
        }
    }
    free(f); free(b); free(lam); free(act);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* parametric_programs.py:442-532 + 349-400 + polyhedron.py:578-604                             */
/* ------------------------------------------------------------------------------------------- */
/*
 * Critical-region table: region r owns rows row_ptr[r] .. row_ptr[r+1]-1 of (A [R x p], b [R]);
 * laws ux [nreg x nu x p], u0 [nreg x nu], Vxx [nreg x p x p], Vx [nreg x p], V0 [nreg].
 * For each point: first region (list order) with max(A x - b) <= tol; idx = -1 if none.
 */
int oracle_point_location(const double *A, const double *b, const int *row_ptr, int nreg, int p, int nu,
                          const double *ux, const double *u0, const double *Vxx, const double *Vx, const double *V0,
                          const double *X, long B, double tol, int *idx, double *u, double *V)
{
    for (long k = 0; k < B; ++k) {
        const double *x = X + (size_t)k * p;
        int found = -1;
        for (int r = 0; r < nreg && found < 0; ++r) {
            double mx = -INFINITY;
            for (int i = row_ptr[r]; i < row_ptr[r + 1]; ++i) {
                double s = 0.0;
                for (int j = 0; j < p; ++j) s += A[(size_t)i * p + j] * x[j];
                s -= b[i];
                if (s > mx) mx = s;
            }
            if (mx <= tol) found = r;
        }
        idx[k] = found;
        if (found >= 0) {
            for (int i = 0; i < nu; ++i) {
                double s = 0.0;
                for (int j = 0; j < p; ++j) s += ux[((size_t)found * nu + i) * p + j] * x[j];
                u[(size_t)k * nu + i] = s + u0[(size_t)found * nu + i];
            }
            double q = 0.0, lin = 0.0;
            for (int i = 0; i < p; ++i) {
                double s = 0.0;
                for (int j = 0; j < p; ++j) s += x[j] * Vxx[((size_t)found * p + j) * p + i];
                q += s * x[i];
                lin += Vx[(size_t)found * p + i] * x[i];
            }
            V[k] = .5 * q + lin + V0[found];
        } else {
            for (int i = 0; i < nu; ++i) u[(size_t)k * nu + i] = NAN;
            V[k] = NAN;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Condensing of the optimal control problem for one mode sequence                              */
/* pympc/dynamics/discrete_time_systems.py:676-751 (condense_pwa_system, productory) and        */
/* pympc/control/controllers.py:682-749 (condense_optimal_control_problem), restated with the   */
/* same dense stacked matrices (Q_bar, R_bar, F_bar, G_bar) the reference forms.                */
/* ------------------------------------------------------------------------------------------- */

/* C[r x c] = A[r x k] B[k x c] */
static void mat_mul(const double *A, const double *B, int r, int k, int c, double *C)
{
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            double s = 0.0;
            for (int q = 0; q < k; ++q) s += A[i * k + q] * B[q * c + j];
            C[i * c + j] = s;
        }
}

/* productory(A_sequence[i:j:-1]) = A_{z(i)} A_{z(i-1)} ... A_{z(j+1)}  (j = -1: down to index 0); identity if empty */
static void productory_desc(const double *As, const int *z, int nx, int i, int j, double *M, double *tmp)
{
    for (int a = 0; a < nx * nx; ++a) M[a] = (a / nx == a % nx) ? 1.0 : 0.0;
    for (int t = i; t > j; --t) { /* left to right: M = M * A_{z(t)} */
        mat_mul(M, As + (size_t)z[t] * nx * nx, nx, nx, nx, tmp);
        memcpy(M, tmp, sizeof(double) * nx * nx);
    }
}

/*
 * modes z[N]; systems As [nm x nx x nx], Bs [nm x nx x nu], cs [nm x nx]; Q, R, P; stage domains of mode i are rows
 * dptr[i] .. dptr[i+1]-1 of DA [. x (nx+nu)], Db [.]; terminal set XA [rN x nx], Xb [rN].
 * Outputs: Abar [(N+1)nx x nx], Bbar [(N+1)nx x N nu], cbar [(N+1)nx], Huu [n x n], Hux [n x nx], Hxx [nx x nx],
 * fu [n], fx [nx], g [1], Au [m x n], Ax [m x nx], b [m],  m = sum_t rows(z[t]) + rN (the caller sizes them).
 */
int oracle_condense(int nx, int nu, int N, int nm, const int *z, const double *As, const double *Bs, const double *cs,
                    const double *Q, const double *R, const double *P, const int *dptr, const double *DA,
                    const double *Db, int rN, const double *XA, const double *Xb, double *Abar, double *Bbar,
                    double *cbar, double *Huu, double *Hux, double *Hxx, double *fu, double *fx, double *g, double *Au,
                    double *Ax, double *b)
{
    const int n = N * nu, nr = (N + 1) * nx;
    for (int t = 0; t < N; ++t)
        if (z[t] < 0 || z[t] >= nm) return -1;
    double *M = (double *)malloc(sizeof(double) * nx * nx), *tmp = (double *)malloc(sizeof(double) * nx * nx);
    double *v = (double *)malloc(sizeof(double) * nx * (nu > 1 ? nu : 1));
    /* free evolution (discrete_time_systems.py:712-713) */
    memset(Abar, 0, sizeof(double) * nr * nx);
    for (int i = 0; i < nx; ++i) Abar[i * nx + i] = 1.0;
    for (int i = 0; i < N; ++i) {
        productory_desc(As, z, nx, i, -1, M, tmp);
        memcpy(Abar + (size_t)(i + 1) * nx * nx, M, sizeof(double) * nx * nx);
    }
    /* forced evolution (:716-721) */
    memset(Bbar, 0, sizeof(double) * nr * n);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j <= i; ++j) {
            if (j < i) {
                productory_desc(As, z, nx, i, j, M, tmp);
                mat_mul(M, Bs + (size_t)z[j] * nx * nu, nx, nx, nu, v);
            } else {
                memcpy(v, Bs + (size_t)z[i] * nx * nu, sizeof(double) * nx * nu);
            }
            for (int r = 0; r < nx; ++r)
                for (int c = 0; c < nu; ++c) Bbar[(size_t)((i + 1) * nx + r) * n + j * nu + c] = v[r * nu + c];
        }
    }
    /* offset term (:724-727) */
    memset(cbar, 0, sizeof(double) * nr);
    for (int i = 0; i < N; ++i) {
        double *ci = cbar + (size_t)(i + 1) * nx;
        for (int j = 0; j < i; ++j) {
            productory_desc(As, z, nx, i, j, M, tmp);
            mat_mul(M, cs + (size_t)z[j] * nx, nx, nx, 1, v);
            for (int r = 0; r < nx; ++r) ci[r] += v[r];
        }
        for (int r = 0; r < nx; ++r) ci[r] += cs[(size_t)z[i] * nx + r];
    }
    free(M); free(tmp); free(v);

    /* stacked cost matrices (controllers.py:716-719) and the blocks of the objective (:722-731) */
    double *Qbar = (double *)calloc((size_t)nr * nr, sizeof(double));
    for (int t = 0; t <= N; ++t) {
        const double *Qt = (t == N) ? P : Q;
        for (int r = 0; r < nx; ++r)
            for (int c = 0; c < nx; ++c) Qbar[(size_t)(t * nx + r) * nr + t * nx + c] = Qt[r * nx + c];
    }
    double *QB = (double *)malloc(sizeof(double) * nr * n), *QA = (double *)malloc(sizeof(double) * nr * nx);
    double *Qc = (double *)malloc(sizeof(double) * nr);
    mat_mul(Qbar, Bbar, nr, nr, n, QB);
    mat_mul(Qbar, Abar, nr, nr, nx, QA);
    mat_mul(Qbar, cbar, nr, nr, 1, Qc);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int q = 0; q < nr; ++q) s += Bbar[(size_t)q * n + i] * QB[(size_t)q * n + j];
            Huu[i * n + j] = s + ((i / nu == j / nu) ? R[(i % nu) * nu + (j % nu)] : 0.0);
        }
        for (int j = 0; j < nx; ++j) {
            double s = 0.0;
            for (int q = 0; q < nr; ++q) s += Bbar[(size_t)q * n + i] * QA[(size_t)q * nx + j];
            Hux[i * nx + j] = s;
        }
        double s = 0.0;
        for (int q = 0; q < nr; ++q) s += Bbar[(size_t)q * n + i] * Qc[q];
        fu[i] = s;
    }
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nx; ++j) {
            double s = 0.0;
            for (int q = 0; q < nr; ++q) s += Abar[(size_t)q * nx + i] * QA[(size_t)q * nx + j];
            Hxx[i * nx + j] = s;
        }
        double s = 0.0;
        for (int q = 0; q < nr; ++q) s += Abar[(size_t)q * nx + i] * Qc[q];
        fx[i] = s;
    }
    {
        double s = 0.0;
        for (int q = 0; q < nr; ++q) s += cbar[q] * Qc[q];
        *g = 0.5 * s;
    }
    free(Qbar); free(QB); free(QA); free(Qc);

    /* stacked constraints (controllers.py:733-747) */
    int m = rN;
    for (int t = 0; t < N; ++t) m += dptr[z[t] + 1] - dptr[z[t]];
    double *Fbar = (double *)calloc((size_t)m * nr, sizeof(double));
    double *Gbar = (double *)calloc((size_t)m * (n > 0 ? n : 1), sizeof(double));
    double *hbar = (double *)malloc(sizeof(double) * (m > 0 ? m : 1));
    int row = 0;
    for (int t = 0; t <= N; ++t) {
        if (t < N) {
            for (int r = dptr[z[t]]; r < dptr[z[t] + 1]; ++r, ++row) {
                for (int c = 0; c < nx; ++c) Fbar[(size_t)row * nr + t * nx + c] = DA[(size_t)r * (nx + nu) + c];
                for (int c = 0; c < nu; ++c) Gbar[(size_t)row * n + t * nu + c] = DA[(size_t)r * (nx + nu) + nx + c];
                hbar[row] = Db[r];
            }
        } else {
            for (int r = 0; r < rN; ++r, ++row) {
                for (int c = 0; c < nx; ++c) Fbar[(size_t)row * nr + N * nx + c] = XA[(size_t)r * nx + c];
                hbar[row] = Xb[r];
            }
        }
    }
    mat_mul(Fbar, Bbar, m, nr, n, Au);
    for (size_t e = 0; e < (size_t)m * n; ++e) Au[e] += Gbar[e];
    mat_mul(Fbar, Abar, m, nr, nx, Ax);
    mat_mul(Fbar, cbar, m, nr, 1, b);
    for (int r = 0; r < m; ++r) b[r] = hbar[r] - b[r];
    free(Fbar); free(Gbar); free(hbar);
    return m;
}
