/*
 * pympc_hip.h -- C ABI of the MI355X (gfx950) batched QP/LP path behind pympc's pnnls backend.
 *
 * The reference (pure Python) has no FFI; the interface every entry point replaces is the backend
 * contract of pympc/optimization/solvers/pnnls.py (same signatures in gurobi.py:5,60 / drake.py:6,95)
 * reached through pympc/optimization/programs.py:9,57.  Each declaration cites the reference lines
 * whose loop it batches.  INTEGRATION.md shows the ctypes stub a pympc maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp64 (or int32/uint32) memory owned by
 *     the caller (torch tensors on the host side); `stream` is a hipStream_t passed as void*.
 *   - entry points allocate nothing and never synchronise; kernels are asynchronous on `stream`.
 *   - return value: 0 on success, < 0 on argument / launch errors (pympc_hip_last_error() has the text).
 *   - per-problem status codes are written to device memory:
 *       PYMPC_OPTIMAL     0   solution written
 *       PYMPC_INFEASIBLE  1   QP infeasible / LP infeasible or unbounded: the reference returns a sol
 *                             dict of None (pnnls.py:128-134, 221-227); outputs are NaN / zero bits
 *       PYMPC_MAXITER     3   NNLS hit 3*cols iterations (scipy raises RuntimeError there)
 *       PYMPC_NOT_PD      4   Hessian not positive definite (numpy.linalg.LinAlgError at pnnls.py:210)
 */
#ifndef PYMPC_HIP_H
#define PYMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYMPC_OPTIMAL 0
#define PYMPC_INFEASIBLE 1
#define PYMPC_MAXITER 3
#define PYMPC_NOT_PD 4

#define PYMPC_ERR_ARG (-1)
#define PYMPC_ERR_UNSUPPORTED (-2)
#define PYMPC_ERR_LAUNCH (-3)

/* ABI version (bumped on any signature change). */
int pympc_hip_abi_version(void);

/* Text of the last error returned by an entry point on this thread. */
const char *pympc_hip_last_error(void);

/*
 * Launch-policy knobs (no counterpart in the reference; for tests and for callers that tune a deployment).  They choose
 * between kernels that implement the SAME method -- never what is computed -- so that every kernel can be driven at every
 * size.  `value` < 0 only queries; `previous` (may be NULL) receives the value before the call.  Keys:
 *   "lp_tile_min_rows" / "lp_tile_min_batch"   pympc_lp_solve_batch runs the eight-LPs-per-workgroup kernel from this many
 *                                              rows / LPs against one shared matrix (defaults 1024 / 1024)
 *   "gram_min_batch"     pympc_mpqp_solve_batch* runs the Gram-form kernel from this many states (default 8192).  NOTE:
 *                        states with 3..8 active constraints are then finished by a different kernel than in a smaller
 *                        batch: same method and decisions, values equal to rounding (1e-9), so argmin / min / multipliers
 *                        of one state can differ in the last bits between a small and a large batch
 *   "learn_min_batch"    certificate-learning pre-pass from this many states (default 131072)
 *   "nnls_itmax_factor"  NNLS iteration limit = factor x columns (default 3 = scipy.optimize.nnls; 0 makes any problem
 *                        that needs a column return PYMPC_MAXITER: how the tests reach that status)
 *   "reserved_cus"       compute units the persistent launches leave free (default 0), for callers whose transfers run
 *                        as copy kernels beside the solve
 *   "dynamic_chunks"     1 (default): with a certificate table and >= 524 288 states, the workgroups of the Gram-form and
 *                        Lawson-Hanson kernels draw chunks of 256 states from a device counter per kernel (in the table's
 *                        header) and share them out to their waves through LDS; 0: a fixed share of interleaved slices
 *                        per workgroup.  Which wave solves a state never changes its result
 *   "prepare_wave"       1 (default): batches of >= 1024 problems with up to 16 variables are factorised one wavefront per
 *                        problem with the constraint rows in registers; 0: the row-tile kernel.  Same sums in the same
 *                        order: bitwise the same workspace
 * Process-wide, read at launch time; not meant to be changed while other threads launch.
 */
int pympc_hip_tune(const char *key, long long value, long long *previous);

/* Largest problem the wave-per-problem QP kernels accept: n <= *n_max variables, m <= *m_max rows. */
void pympc_hip_limits(int *n_max, int *m_max);

/* -------------------------------------------------------------------------------------------------
 * Shared-matrix batched QP: MultiParametricQuadraticProgram.solve(x) for a batch of states.
 * Reference: pympc/optimization/parametric_programs.py:155-179 (f = Hux x + fu, b = b - Ax x,
 * quadratic_program, lift of min) -> pympc/optimization/solvers/pnnls.py:149-241, looped by
 * pympc/control/controllers.py:76-124 (feedforward / feedback).
 * ------------------------------------------------------------------------------------------------- */

/* Number of doubles of caller-owned device workspace `ws` that pympc_mpqp_prepare fills (factorised constant part +
 * the Gram matrix of its rows, mp x mp doubles with mp = m rounded up to the kernel's column padding: 1.2 MB for m = 370). */
size_t pympc_mpqp_workspace_doubles(int n, int m, int p);

/*
 * One-time factorisation of the constant part (hoists pnnls.py:210-214 out of the per-state loop):
 * L = chol(Huu), L^-1, Huu^-1, M = Au L^-T, and the affine maps m(x) = b(x) + Au Huu^-1 f(x),
 * u_unc(x) = -Huu^-1 f(x).   Huu [n x n], Hux [n x p], Hxx [p x p], fu [n], fx [p], g scalar (host value),
 * Au [m x n], Ax [m x p], b [m].   ws_status (1 int32, device) receives PYMPC_OPTIMAL or PYMPC_NOT_PD.
 */
int pympc_mpqp_prepare(int n, int m, int p, const double *Huu, const double *Hux, const double *Hxx, const double *fu,
                       const double *fx, double g, const double *Au, const double *Ax, const double *b, double *ws,
                       int32_t *ws_status, void *stream);

/*
 * Solve the QP at each of the B states X [B x p].  Outputs (any pointer except status may be NULL):
 *   argmin  [B x n_out]   first n_out entries of the minimiser (n_out = n for feedforward, nu for feedback)
 *   vmin    [B]           lifted optimal value (parametric_programs.py:177)
 *   mult    [B x m]       multiplier_inequality (pnnls.py:232)
 *   active  [B x ceil(m/32)] uint32 bit j of word j/32 set iff multiplier j > tol (pnnls.py:235)
 *           (in a launch of >= 8192 states the kernels also use a state's active words as hand-over space while it is
 *           pending: a state the Gram-form kernel leaves at 8 passive columns starts the Lawson-Hanson kernel from those
 *           columns when `active` is given and from nothing when it is NULL -- the same result to rounding (1e-8), the
 *           same decisions, but not the same last bits)
 *   status  [B]           PYMPC_OPTIMAL / PYMPC_INFEASIBLE / PYMPC_MAXITER
 *   iters   [B x 2]       (outer, inner) NNLS iteration counters (roofline accounting)
 */
int pympc_mpqp_solve_batch(int n, int m, int p, const double *ws, const double *X, int64_t B, double tol, int n_out,
                           double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status,
                           int32_t *iters, void *stream);

/*
 * The same solve with a table of learned infeasibility certificates (no counterpart in the reference, which
 * solves every state from scratch at pnnls.py:216-227).  When the NNLS of a state ends with a zero residual
 * (pnnls.py:230, "infeasible") its solution y >= 0 is a Farkas certificate M'y = 0, m(x)'y < 0 whose first part
 * does not depend on x: it proves infeasibility of every state x' in the half-space y'm0 + (Gm y)'x' < 0.  The
 * kernel stores up to 256 verified certificates in `cuts` and rejects later states that violate one of them by a
 * relative margin of 1e-6 without iterating (status PYMPC_INFEASIBLE, iters (0, 0)); states nearer to the boundary
 * take the full solve, so the feasibility decision is the reference's.  `cuts`: pympc_mpqp_cuts_doubles(p) doubles
 * of device memory owned by the caller, zeroed by pympc_mpqp_cuts_reset before first use and whenever `ws` is
 * prepared again; it belongs to ONE prepared mpQP, is read and appended to by the kernel and can be kept across calls.
 * Launches that share a table must be ordered on ONE stream (the screening kernel reads the entries that were complete
 * when it started through the scalar cache, and the table's header holds the chunk counters of the launch's kernels; a
 * concurrent launch on the same table is not supported).
 * NULL, p = 0 or p > 8: no certificates (plain solve).
 */
size_t pympc_mpqp_cuts_doubles(int p);
int pympc_mpqp_cuts_reset(int p, double *cuts, void *stream);

/*
 * Certificates at set-up time (no counterpart in the reference).  The certificates of an mpQP are half-spaces that
 * contain no feasible parameter: facets (or valid combinations of facets) of ONE polytope, the feasible set of
 * parametric_programs.py:243-260 (get_feasible_set).  pympc_mpqp_cuts_learn solves the B sample states X [B x p] with
 * the Lawson-Hanson kernel alone, writes nothing but status [B] (scratch) and appends the certificates of the infeasible
 * ones to `cuts`; pympc_mpqp_cuts_mark declares the entries stored so far part of the prepared mpQP, and
 * pympc_mpqp_cuts_restore drops everything learned since the mark (pympc_mpqp_cuts_reset still empties the table,
 * mark included).  The host layer samples the bounding box of the feasible set at several scales right after
 * pympc_mpqp_prepare, so that the first large batch starts with the table a long-running caller would have.
 */
int pympc_mpqp_cuts_learn(int n, int m, int p, const double *ws, double *cuts, const double *X, int64_t B, double tol,
                          int32_t *status, void *stream);
int pympc_mpqp_cuts_mark(int p, double *cuts, void *stream);
int pympc_mpqp_cuts_restore(int p, double *cuts, void *stream);
int pympc_mpqp_solve_batch_cuts(int n, int m, int p, const double *ws, double *cuts, const double *X, int64_t B, double tol,
                                int n_out, double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status,
                                int32_t *iters, void *stream);

/*
 * pympc_mpqp_solve_batch_cuts for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The certificate table
 * lives in a file-scope device array of the library (18 816 bytes per GPU, held until the process ends; its device
 * address is looked up once per GPU with hipGetSymbolAddress on the first call there) instead of caller memory, and
 * is emptied at the start of EVERY call (hipMemsetAsync on `stream`), because the certificates of the last call belong
 * to that call's mpQP.  Needs 1 <= p <= 8 (the certificates' limit): anything else returns PYMPC_ERR_UNSUPPORTED.
 * Calls neither hipMalloc nor hipFree.  All calls on one GPU share that GPU's table: calls that overlap in time on one
 * GPU (different streams or threads) race on it and must be serialised by the caller.  Outputs as
 * pympc_mpqp_solve_batch_cuts with an empty table (statuses and solutions identical to the plain solve).
 */
int pympc_mpqp_solve_batch_static(int n, int m, int p, const double *ws, const double *X, int64_t B, double tol, int n_out,
                                  double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status,
                                  int32_t *iters, void *stream);

/* -------------------------------------------------------------------------------------------------
 * Per-problem batched QP: B independent problems  min .5 x'H_k x + f_k'x  s.t. A_k x <= b_k  of equal
 * shape (n, m); equalities are folded by the caller as rows (C, -C) exactly as pnnls.py:202-207 does.
 * Reference: pympc/optimization/solvers/pnnls.py:149-241 once per problem (programs.py:57-105); callers
 * polyhedron.py:523 (empty), controllers.py:668-680 + parametric_programs.py:173 (get_mpqp(ms).solve).
 * H [B x n x n] (or [n x n] with h_stride 0), f [B x n], A [B x m x n] (a_stride 0: shared), b [B x m].
 * n <= 63: one wavefront per problem (pympc_hip_limits); 64 <= n <= 1023 or 2048 < m <= 32768: one workgroup per problem with
 * the factors in device memory (csrc/qp_block.hpp) -- same outputs, same statuses (PYMPC_NOT_PD included).
 * ------------------------------------------------------------------------------------------------- */
size_t pympc_qp_workspace_doubles(int n, int m, int64_t B);

int pympc_qp_solve_batch(int n, int m, int64_t B, const double *H, int64_t h_stride, const double *f, const double *A,
                         int64_t a_stride, const double *b, double tol, double *ws, double *argmin, double *vmin,
                         double *mult, uint32_t *active, int32_t *status, int32_t *iters, void *stream);

/* -------------------------------------------------------------------------------------------------
 * B different mpQPs of equal shape (n, m, p), each solved at its own parameter (x_stride = p) or all at one
 * shared parameter (x_stride = 0): HybridModelPredictiveController.get_mpqp(mode_sequence).solve(x0) over a batch
 * of mode sequences.
 * Reference: pympc/control/controllers.py:668-680 (get_mpqp) -> pympc/optimization/parametric_programs.py:155-179
 * (solve) -> pympc/optimization/solvers/pnnls.py:149-241, once per mode sequence.
 * Huu [B x n x n], Hux [B x n x p], Hxx [B x p x p], fu [B x n], fx [B x p], g [B], Au [B x m x n], Ax [B x m x p],
 * b [B x m] (the arrays pympc_condense_batch writes); X [B x p] or [p].  ws: pympc_mpqp_multi_workspace_doubles(...)
 * doubles of device scratch.  Outputs as pympc_mpqp_solve_batch with n_out = n.
 * ------------------------------------------------------------------------------------------------- */
size_t pympc_mpqp_multi_workspace_doubles(int n, int m, int p, int64_t B);

int pympc_mpqp_multi_solve(int n, int m, int p, int64_t B, const double *Huu, const double *Hux, const double *Hxx,
                           const double *fu, const double *fx, const double *g, const double *Au, const double *Ax,
                           const double *b, const double *X, int64_t x_stride, double tol, double *ws, double *argmin,
                           double *vmin, double *mult, uint32_t *active, int32_t *status, int32_t *iters, void *stream);

/* -------------------------------------------------------------------------------------------------
 * Batched condensing of the optimal control problem of a PWA system: one mpQP per mode sequence (fp64 MFMA).
 * Reference: pympc/control/controllers.py:682-749 (condense_optimal_control_problem) and
 * pympc/dynamics/discrete_time_systems.py:676-751 (condense_pwa_system, productory), called by
 * controllers.py:58-74 (ModelPredictiveController._condense_program) and :668-680 (get_mpqp).
 * modes [B x N] int32 in [0, nm); As [nm x nx x nx], Bs [nm x nx x nu], cs [nm x nx] (c = 0 for a linear system);
 * Q [nx x nx], R [nu x nu], P [nx x nx]; stage domains F x + G u <= h of every mode padded by the caller to rmax
 * rows with F = 0, G = 0, h = 1: Fd [nm x rmax x nx], Gd [nm x rmax x nu], hd [nm x rmax]; terminal set
 * FN [rN x nx], hN [rN].  With n = N nu, p = nx, m = N rmax + rN the outputs are Huu [B x n x n], Hux [B x n x p],
 * Hxx [B x p x p], fu [B x n], fx [B x p], g [B], Au [B x m x n], Ax [B x m x p], b [B x m], optionally
 * Sbar [B x (N+1) nx x (nx + n + 1)] = [A_bar | B_bar | c_bar] (discrete_time_systems.py:729; may be NULL) and
 * status [B] (0, or PYMPC_BAD_MODE when a mode index is out of range: nothing else is written for that problem).
 * ------------------------------------------------------------------------------------------------- */
#define PYMPC_BAD_MODE 5

/* Largest system the condensing kernel accepts: nx <= *nx_max, nx + N nu + 1 <= *w_max. */
void pympc_condense_limits(int *nx_max, int *w_max);

int pympc_condense_batch(int nx, int nu, int N, int nm, int rmax, int rN, const int32_t *modes, int64_t B,
                         const double *As, const double *Bs, const double *cs, const double *Q, const double *R,
                         const double *P, const double *Fd, const double *Gd, const double *hd, const double *FN,
                         const double *hN, double *Huu, double *Hux, double *Hxx, double *fu, double *fx, double *g,
                         double *Au, double *Ax, double *b, double *Sbar, int32_t *status, void *stream);

/* -------------------------------------------------------------------------------------------------
 * Batched LP: B problems  min f_k'x  s.t.  A_k x <= b_k  of equal shape (n, m); equalities are folded by the
 * caller as rows (C, -C) exactly as pnnls.py:101-106 does.
 * Reference: pympc/optimization/solvers/pnnls.py:41-147 (linear_program) + :5-39 (pnnls) once per problem
 * (programs.py:9-55); the loops it batches are pympc/geometry/polyhedron.py:435-448 (minimal_facets: one LP per
 * facet against a SHARED constraint matrix with the i-th right-hand side relaxed by 1, :439-441) and
 * polyhedron.py:751-756 (Chebyshev centre).
 * f [B x n]; A [B x m x n] (a_stride = m*n) or [m x n] shared (a_stride 0); b [B x m] (b_stride = m) or [m]
 * shared (b_stride 0); relax_idx [B] int32 or NULL: b_k[relax_idx[k]] += relax_val (a negative index relaxes
 * nothing).  ws: pympc_lp_workspace_doubles(...) doubles of device scratch (transposed matrix + row norms).
 * Outputs (any pointer except status may be NULL): argmin [B x n], vmin [B], mult [B x m] (multiplier_inequality,
 * pnnls.py:140), active [B x ceil(m/32)] bits of mult > tol (pnnls.py:141), status [B] (PYMPC_OPTIMAL /
 * PYMPC_INFEASIBLE = infeasible or unbounded, every output None in the reference / PYMPC_MAXITER),
 * iters [B x 3] (phase-1 outer, phase-1 inner, phase-2 pivots).
 * ------------------------------------------------------------------------------------------------- */
size_t pympc_lp_workspace_doubles(int n, int m, int64_t B, int a_shared);

/* Largest LP the workgroup-per-problem kernel accepts. */
void pympc_lp_limits(int *n_max, int *m_max);

int pympc_lp_solve_batch(int n, int m, int64_t B, const double *f, const double *A, int64_t a_stride, const double *b,
                         int64_t b_stride, const int32_t *relax_idx, double relax_val, double tol, double *ws,
                         double *argmin, double *vmin, double *mult, uint32_t *active, int32_t *status, int32_t *iters,
                         void *stream);

/* -------------------------------------------------------------------------------------------------
 * Raw NNLS and PNNLS: the two numerical cores under the QP / LP transformations.
 *
 * pympc_nnls_batch: B problems  min |A_k x - b_k|  s.t. x >= 0  of equal shape (rows <= 1024, cols <= 32768; up to
 * 64 rows one wavefront per problem with a row per lane, from 65 rows one workgroup per problem with a row per
 * thread, csrc/nnls_block.hpp -- same algorithm, same selection rules).
 * Reference: scipy.optimize.nnls(A, b) as called at pympc/optimization/solvers/pnnls.py:36 and :218 (third party,
 * pinned scipy<=1.11.4: Lawson-Hanson; maxiter = 3 cols).  A [B x rows x cols] (a_stride = rows*cols) or [rows x cols]
 * shared (a_stride 0); b [B x rows].  Outputs x [B x cols], rnorm [B] = |A x - b| (may be NULL), status [B]
 * (PYMPC_OPTIMAL / PYMPC_MAXITER: scipy raises RuntimeError there), iters [B x 2] (may be NULL).
 *
 * pympc_pnnls_batch: B problems  min |A_k v + B_k u - c_k|  s.t. v >= 0  (rows <= 1024, cols <= 32768, nu <= 256; up to 64 of each on the
 * wave kernel, beyond on the workgroup kernels of csrc/nnls_block.hpp, B in LDS while rows * nu * 8 bytes fit and nu <= 64).
 * Reference: pnnls(A, B, c) of pympc/optimization/solvers/pnnls.py:5-39 (returns v, u, r); the reference projects
 * with the dense matrix I - B pinv(B) (:30-33), here u is eliminated by a column-pivoted Householder QR of B (same
 * residual; u is the minimum-norm solution when B is rank deficient, as pinv gives).  A [B x rows x cols],
 * Bm [B x rows x nu], c [B x rows].  Outputs v [B x cols], u [B x nu], rnorm [B] (may be NULL), status, iters as above.
 * ws: *_workspace_doubles(...) doubles of device scratch.
 * ------------------------------------------------------------------------------------------------- */
size_t pympc_nnls_workspace_doubles(int rows, int cols, int64_t B);
int pympc_nnls_batch(int rows, int cols, int64_t B, const double *A, int64_t a_stride, const double *b, double *ws, double *x,
                     double *rnorm, int32_t *status, int32_t *iters, void *stream);
size_t pympc_pnnls_workspace_doubles(int rows, int cols, int nu, int64_t B);
int pympc_pnnls_batch(int rows, int cols, int nu, int64_t B, const double *A, const double *Bm, const double *c, double *ws,
                      double *v, double *u, double *rnorm, int32_t *status, int32_t *iters, void *stream);

/* -------------------------------------------------------------------------------------------------
 * Explicit-MPC point location + affine law: first critical region (list order) containing each point.
 * Reference: pympc/optimization/parametric_programs.py:442-463 (get_critical_region), :465-532 (u, V),
 * :349-400 (CriticalRegion.u / V), pympc/geometry/polyhedron.py:578-604 (contains, tol 1e-7).
 * Table: region r owns rows row_ptr[r] .. row_ptr[r+1]-1 of A [total_rows x p], b [total_rows]
 * (total_rows = row_ptr[nreg], passed by the host so that nothing is read back); laws ux [nreg x nu x p],
 * u0 [nreg x nu] (first nu rows of the region's law), Vxx [nreg x p x p], Vx [nreg x p], V0 [nreg].
 * box [nreg x 2 p] (p lower bounds, then p upper bounds; may be NULL): per region an axis-aligned box that contains
 * every point its row test can accept; regions whose box holds none of a wavefront's points are skipped.  It only
 * prunes: the result is the reference's first-hit region whatever the boxes are, as long as they are not too small.
 * Cell grid (may be NULL / 0): grid_geom [4 p] = lower corner of a uniform grid of grid_n cells per axis, then
 * 1 / cell size per axis, then the upper and the lower corner of the box outside which a point lies in no region (used
 * with grid_sure only; +-inf switches that shortcut off); grid_masks [grid_n^p x ceil(nreg/64)] uint64, cell index = ((c_0 grid_n + c_1) grid_n + ...),
 * bit r % 64 of word r / 64 set iff region r can contain a point of the cell (points outside the grid are looked up
 * in the nearest cell).  Like the boxes it only prunes: a missing bit is only allowed for a region that contains no
 * point of the cell.
 * grid_sure [grid_n^p] int32 (may be NULL; needs the grid): r >= 0 = every point of the cell passes the row test of
 * region r for any tol >= 0 and no earlier region can contain a point of the cell; -1 = no region can; -2 = undecided
 * (the cell's bit set is scanned).  Points outside the box of grid_geom are taken to lie in no region; the others are
 * looked up in the nearest cell, so with an unbounded box the border cells must be -2.  Unlike the boxes and bit sets
 * this one DECIDES: a wrong entry is a wrong result.
 * grid_lists (int32 pool, may be NULL): grid_sure <= -3 means -3 - grid_sure is the offset of the cell's decision list
 * [ncand, (region, nrows, row_0 .. row_{nrows-1}) x ncand]: the candidate regions of the cell in ascending order, each
 * with the rows (indices into A / b) that do not hold on the whole cell; a point of the cell lies in the first
 * candidate whose listed rows it satisfies (A x - b <= tol, as everywhere), in no region if there is none.  Without the
 * pool such cells count as undecided.  grid_scan: 0 = the caller guarantees that no cell is -2 (the scan pass over the
 * marked points is not launched); anything else = launch it.
 * Outputs: idx [B] int32 (-1 when uncovered), u [B x nu] (NaN when uncovered), V [B] (NaN when uncovered).
 * ------------------------------------------------------------------------------------------------- */
int pympc_point_location(int nreg, int total_rows, int p, int nu, const double *A, const double *b,
                         const int32_t *row_ptr, const double *ux, const double *u0, const double *Vxx,
                         const double *Vx, const double *V0, const double *box, const double *grid_geom, int grid_n,
                         const uint64_t *grid_masks, const int32_t *grid_sure, const int32_t *grid_lists, int grid_scan,
                         const double *X, int64_t B, double tol, int32_t *idx, double *u, double *V, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PYMPC_HIP_H */
