/*
 * mvf.h -- C ABI of libmvf.so, the MI355X (gfx950) morphometric vector-field engine.
 *
 * This is the drop-in boundary for ONE hot path of aristoteleo/spateo-release: the SparseVFC kernel regression
 * under spateo.tdr (con_K -> EM loop -> coefficient solve -> differential-geometry evaluators).  The reference has
 * no native code and no FFI for this path (SURVEY.md "Three facts", section 8b): its arithmetic lives in Python
 * (third-party `dynamo` + in-tree twins).  Each entry point below therefore cites the reference *Python* interface
 * it replaces (file:line under /root/reference); INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.  All data pointers are DEVICE pointers
 *     (hipMalloc'd by the caller, e.g. torch tensors' data_ptr()); the library never allocates or frees memory
 *     and never takes ownership.  Workspaces are caller-provided and sized by the *_workspace_bytes queries.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call is asynchronous on it.
 *   - `dtype` selects the arithmetic of the cell-sized arrays: MVF_F32 or MVF_F64.  Reductions, the Gram matrix,
 *     the coefficient solve and the coefficients C are always float64.
 *   - Row-major, dense, no strides.  Cell coordinates / displacements use the padded layout "x4": n rows of
 *     4 elements (x, y, z, 0) so that one lane loads one cell with a single 16/32-byte access; 2-D data sets z = 0.
 *   - Return value: 0 = ok; non-zero = error, message in mvf_last_error() (thread-local).  No exceptions cross
 *     the boundary.  The Python host raises RuntimeError on a non-zero status.
 *   - State: the library has no global or per-process state (mvf_last_error's thread-local buffer, a per-device cache
 *     of the CU count and the developer options of mvf_debug_option - all at their defaults unless that entry point is
 *     called - aside), and it never reads the environment.  Whatever must survive between calls lives in CALLER-provided memory and is named at
 *     the entry point that uses it: the workspace of mvf_solve_minnorm_lr keeps the pivot order of its last finished
 *     factorisation (the next call's hint), the workspaces of the two minimum-norm solves keep their decomposition for
 *     `reuse` calls and mvf_pinv_diag, `basis` keeps the eigenvectors of mvf_solve_minnorm for its warm start.  A fresh
 *     (or overwritten) workspace simply means no hint / no reuse.  The one opaque object is the `mvf_comm` handle of the
 *     multi-GPU exchange (an RCCL communicator, created and destroyed by the caller; libmvf.so links librccl for it).
 */
#ifndef MVF_H
#define MVF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MVF_F32 = 0, MVF_F64 = 1 } mvf_dtype;

/* evaluator output selection flags for mvf_eval */
enum {
    MVF_EVAL_V = 1,        /* v(x)                      n x 3            */
    MVF_EVAL_JAC = 2,      /* Jacobian                  (3, 3, n) layout */
    MVF_EVAL_DIV = 4,      /* divergence = trace J      n                */
    MVF_EVAL_CURL = 8,     /* curl 3-vector             n x 3            */
    MVF_EVAL_ACC = 16,     /* acceleration a = J v      n x 3            */
    MVF_EVAL_CURV = 32,    /* curvature vector (f. 2)   n x 3            */
    MVF_EVAL_TORS = 64,    /* torsion 3-vector          n x 3            */
    MVF_EVAL_JDET = 128    /* det J                     n                */
};

/* ---- library ------------------------------------------------------------------------------------------------ */
const char* mvf_last_error(void);
int mvf_version(void);                 /* ABI version, currently 7 */
/* mvf_read_back (ABI 7): blocking device -> host copy of a small status block on `stream` (everything enqueued on the stream
 * before it is complete when it returns) - the one host read of an EM iteration (statistics, solver status, sum P r). */
int mvf_read_back(void* dst_host, const void* src_device, size_t nbytes, void* stream);
/* Developer options, process-wide: which kernel variant / launch plan is taken in A/B measurements and in the tests that
 * compare the variants bit for bit.  The library NEVER reads the environment (rounds 1 - 3 had getenv knobs in launch
 * paths); nothing but this call changes its behaviour.  value 0 = default.  Nine names (round 6 removed four that no test
 * or measurement used any more): "conk_form" (1 rows, 2 flat, 3 2d), "slice_len" (cells per Gram slice), "solve_small_off"
 * (1: the blocked multi-launch Cholesky at every m), "lr_no_deflate" (1: mvf_solve_minnorm_lrd always takes the Jacobi path),
 * "defl_block" (64 / 128 / 256: mvf_solve_minnorm_lrd tries that block size alone), "defl_apps" (1 .. 8: applications of
 * S2^-1 in its block inverse iteration), "lr_no_direct" (1: mvf_solve_minnorm_lrd never takes its direct form),
 * "direct_accept" (v > 0: the direct form accepts at most v - 1 deflated directions), "lr_timing" (1: phase times of
 * mvf_solve_minnorm_lr on stderr).  Unknown name: non-zero return.  mvf_debug_option_get returns -1 for an unknown name. */
int mvf_debug_option(const char* name, long long value);
long long mvf_debug_option_get(const char* name);
int mvf_device_count(int* count);      /* number of visible HIP devices (0 without a GPU) */

/* ---- preprocessing: sorted unique rows -------------------------------------------------------------------------------
 * Replaces: `tmp_X, uid = np.unique(X, axis=0, return_index=True)` of dynamo SparseVFC (same
 * call in-tree: spateo/alignment/methods/morpho_class.py:845).  X: n x d float64 row-major (finite).  Outputs (device,
 * caller-allocated for n rows): uid[0..count) = index of the FIRST occurrence of each distinct row, in lexicographic row
 * order; rows[0..count) = those rows; count[0] = number of distinct rows.  Stable LSD radix sort over the columns
 * (hand-written from round 4 on: 8-bit passes of per-chunk histograms, a table scan and a ballot-ranked stable scatter;
 * rounds 2 - 3 called rocPRIM here) + run flags + a ballot-ranked compaction; bit-identical to NumPy for finite input. */
size_t mvf_unique_rows_workspace_bytes(int64_t n, int d);
int mvf_unique_rows(const double* X, int64_t n, int d, int64_t* uid, double* rows, int64_t* count, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ---- preprocessing: kNN bandwidth -------------------------------------------------------------------------------------
 * Replaces the neighbour search of dynamo `bandwidth_selector` (exact kNN with
 * k = max(2, int(0.2 m)) neighbours incl. the point itself, `d = mean(dist[:, 1:]) / 1.5`, `h = sqrt(2) d`).
 * X: m x d float64 row-major (device), 1 <= m <= 8192, d <= 8.  rowsum[i] (device, m float64) = sum of the distances from
 * point i to its k - 1 nearest OTHER points; the caller takes mean = sum(rowsum) / (m (k - 1)).  One workgroup per point:
 * all m squared distances in LDS, bitonic sort, fixed-order sum (deterministic). */
int mvf_knn_rowsum(const double* X, int64_t m, int d, int k, double* rowsum, void* stream);

/* ---- grid preparation: convex-hull mask --------------------------------------------------------------------------------
 * Replaces: `grid_in_hull = in_hull(Grid, hull.points[hull.vertices, :])` of `get_X_Y_grid`
 * (spateo/tdr/interpolations/utils.py:48-53; `in_hull` = Delaunay(...).find_simplex(p) >= 0, spateo/tools/utils.py:205-221).
 * A point lies in a convex hull iff it is on the inner side of every facet: inside[i] = (max_f n_f . p_i + d_f <= tol).
 * points: n x 3 float64 row-major; equations: nfacets x 4 float64 = SciPy `ConvexHull(X).equations` (built on the host
 * with Qhull, as the reference does); inside: n bytes (0 / 1).  tol: the caller passes 100 eps x the hull's extent,
 * the scale of find_simplex's own barycentric tolerance. */
int mvf_hull_mask(const double* points, int64_t n, const double* equations, int64_t nfacets, double tol,
                  unsigned char* inside, void* stream);

/* ---- con_K ----------------------------------------------------------------------------------------------------
 * K[i, j] = exp(-beta * ||x_i - y_j||^2), materialised n x m row-major.
 * Replaces: dynamo `con_K` / in-tree twin `_con_K` spateo/tdr/morphometrics/morphofield/gaussian_process.py:16-36
 * (cdist path :21-24).  x: n x d, y: m x d, plain row-major (NOT x4), 1 <= d <= 8.  HBM-write-bound. */
int mvf_con_k(const void* x, int64_t n, const void* y, int64_t m, int d, double beta, void* K, mvf_dtype dtype,
              void* stream);
/* return_d=True variant (gaussian_process.py:25-29): also D[n, :, m] = x_n - y_m as n x d x m. */
int mvf_con_k_d(const void* x, int64_t n, const void* y, int64_t m, int d, double beta, void* K, void* D,
                mvf_dtype dtype, void* stream);

/* ---- field evaluation:  V = con_K(x, ctrl, beta) @ C  ---------------------------------------------------------
 * Replaces: dynamo `vector_field_function` (call site differential_geometry.py:67-68; twin `_gp_velocity`
 * gaussian_process.py:102-127), `V = U.dot(C)` and `grid_V = grid_U.dot(C)` of SparseVFC.
 * U is never materialised: kernel values are recomputed from x and ctrl.  x4: n x 4, ctrl4: m x 4 (dtype),
 * C: m x 3 float64 (unused columns zero), V4 out: n x 4 (dtype, 4th = 0).
 * If y4 != NULL also writes r[i] = ||y_i - V_i||^2 (dtype) and, if P != NULL (dtype, n), accumulates
 * stats[0] += sum_i P_i r_i  (float64; caller zeroes stats).  Reductions over cells are DETERMINISTIC everywhere in
 * this library: per-workgroup partials go to `scratch` (caller-provided, >= mvf_reduce_scratch_doubles(n) float64,
 * needed whenever P != NULL) and are summed in workgroup order by one workgroup - no floating-point atomics. */
size_t mvf_reduce_scratch_doubles(int64_t n);
int mvf_apply(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C, void* V4,
              const void* y4, const void* P, void* r, double* stats, double* scratch, mvf_dtype dtype, void* stream);

/* ---- E-step ---------------------------------------------------------------------------------------------------
 * Replaces: dynamo `get_P` + the `P = max(P, minP)` / numcorr lines of SparseVFC.
 * Input r (dtype, n) = ||Y - V||^2.  Two phases so that the min-non-zero rule `temp1[temp1 == 0] =
 * min(temp1[temp1 != 0])` can be made global across ranks between them:
 *   mvf_estep_min : mins[0] = min non-zero exp(-r/(2 sigma2)) (float64, +inf if none), mins[1] = #zeros;
 *                   `mins` must hold MVF_ESTEP_MIN_DOUBLES float64 (the tail is block-partial scratch)
 *   mvf_estep_p   : P_out (dtype, n) = max(P, minP) with P = t1/(t1+t2);  stats (float64[5], caller zeroes):
 *                   [0] += sum P_unfloored * r, [1] += sum P_unfloored, [2] += sum P_floored, [3] += #(P_floored > theta),
 *                   [4] += #(t1 == 0), the cells that took the fill.  The fill is t1_zero_fill_dev[0] when that
 *                   DEVICE pointer is non-NULL (mins[0] of mvf_estep_min, MIN-all-reduced across ranks by the host; +inf
 *                   -> 0): the two phases then chain on the stream with no host round trip; else t1_zero_fill.
 *                   scratch: >= mvf_reduce_scratch_doubles(n) float64 (deterministic two-level sums).
 * All arithmetic in float64 regardless of dtype. */
#define MVF_ESTEP_MIN_DOUBLES 4098
int mvf_estep_min(const void* r, int64_t n, double sigma2, double* mins, mvf_dtype dtype, void* stream);
int mvf_estep_p(const void* r, int64_t n, double sigma2, double gamma, double a, int dy, double minP, double theta,
                double t1_zero_fill, const double* t1_zero_fill_dev, void* P_out, double* stats, double* scratch,
                mvf_dtype dtype, void* stream);
/* mvf_estep (ABI 7): mvf_estep_min followed by mvf_estep_p with the fill taken from mins[0] on the device - the E-step of ONE
 * process (no MIN all-reduce between the phases) in one call.  `stats` is OVERWRITTEN (no memset by the caller).  n >= 1. */
int mvf_estep(const void* r, int64_t n, double sigma2, double gamma, double a, int dy, double minP, double theta, double* mins,
              void* P_out, double* stats, double* scratch, mvf_dtype dtype, void* stream);

/* ---- M-step assembly:  G = U^T diag(P) U (m x m),  R = U^T diag(P) Y (m x 3)  --------------------------------
 * Replaces: `UP = U.T * repmat(P.T, M, 1); lhs = UP.dot(U) ...; rhs = UP.dot(Y)` of SparseVFC (App. A 5c; same
 * shape in-tree at spateo/alignment/methods/morpho_class.py:1266-1293).  MFMA kernel; the kernel values (cell dtype)
 * are regenerated from x4/ctrl4 in registers and accumulated with v_mfma_f64_16x16x4_f64 in both dtypes, so G is the
 * exact Gram matrix of those values; per-slice partial tiles are summed in a fixed order (deterministic).  No
 * process-global state: the library's behaviour depends on its arguments only.
 * Outputs are float64: G (m x m, full symmetric), R (m x 3).  They hold THIS rank's partial sums; the exchange is
 * mvf_allreduce_stats (below) or the host's own collective on the same device pointers: the packed triangle of G is
 * all-reduced asynchronously while the rhs kernels run, then [R | scalars]. */
size_t mvf_gram_workspace_bytes(int64_t n, int64_t m, mvf_dtype dtype);
int mvf_gram(const void* x4, const void* P, const void* y4, int64_t n, const void* ctrl4, int64_t m, double beta,
             double* G, double* R, void* workspace, size_t workspace_bytes, mvf_dtype dtype, void* stream);
/* Same, one stage at a time (mask of MVF_GRAM_STAGE_*): TILES = the MFMA kernel writing per-slice partial tiles into
 * the workspace, RHS = the U^T P Y kernel (partials), REDUCE / REDUCE_RHS = fixed-order sums of the partials into G /
 * into R.  mvf_gram == all four.  Lets a caller bracket the dominant kernel with HIP events on `stream` (bench.py's
 * roofline) and lets a wide Y (Dy > 3, kernel_interpolation) reuse one G for several 3-column rhs groups. */
enum { MVF_GRAM_STAGE_TILES = 1, MVF_GRAM_STAGE_RHS = 2, MVF_GRAM_STAGE_REDUCE = 4, MVF_GRAM_STAGE_REDUCE_RHS = 8 };
int mvf_gram_stages(int stages, const void* x4, const void* P, const void* y4, int64_t n, const void* ctrl4,
                    int64_t m, double beta, double* G, double* R, void* workspace, size_t workspace_bytes,
                    mvf_dtype dtype, void* stream);

/* Cached-U variant of the Gram kernel.  U = con_K(x, ctrl) is constant across EM iterations (only P changes), so when
 * HBM has room (mvf_ublk_bytes = sizeof(dtype) * roundup(n,256) * roundup(m,128) bytes; 98 GB at 8 M x 3000 float32)
 * the kernel values are materialised once per fit in the MFMA-operand-shaped layout Ublk[m/16][n][16] and the Gram
 * kernel streams them (9 coalesced loads per 16 MFMAs) instead of regenerating them - VALU work is additive to f64
 * MFMA time on gfx950, and in float64 mode the regenerated exp alone makes the kernel VALU-bound.  Same values, same
 * float64 accumulation, bit-identical outputs to mvf_gram; `stages` as in mvf_gram_stages. */
size_t mvf_ublk_bytes(int64_t n, int64_t m, mvf_dtype dtype);
int mvf_ublk_build(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, void* ublk, size_t ublk_bytes,
                   mvf_dtype dtype, void* stream);
int mvf_gram_cached(int stages, const void* ublk, const void* x4, const void* P, const void* y4, int64_t n,
                    const void* ctrl4, int64_t m, double beta, double* G, double* R, void* workspace,
                    size_t workspace_bytes, mvf_dtype dtype, void* stream);

/* ---- wide right-hand sides (Dy > 3) on the cached kernel values (ABI 6) -------------------------------------------
 * Replaces: dynamo's `rhs = UP.dot(Y)` and `V = U.dot(C)` when Y has many columns - what
 * `kernel_interpolation` passes (spateo/tdr/interpolations/interpolation_sparseVFC.py:13-85: Y = the expression of Dy genes).
 * Both stream the cache of mvf_ublk_build ONCE for all columns, as v_mfma_f64_16x16x4_f64 products:
 *   mvf_rhs_cached  : R[j][d] = sum_n U[n][j] P_n Y[n][d]        R: m x dy float64, leading dimension ldr >= dy
 *   mvf_apply_cached: V[n][d] = sum_j U[n][j] C[j][d],  r_n = sum_d (Y[n][d] - V[n][d])^2 (against V as stored),
 *                     stats[0] += sum_n P_n r_n (P may be NULL: no sum)
 * Yd / Vd: n rows x ldy columns of the cell dtype, row-major, ldy a multiple of 16 and >= dy; Yd must be READABLE for
 * mvf_ublk_npad rows (n rounded up to 256) and zero (or any finite value) beyond row n and column dy.  C: float64,
 * leading dimension ldc (a multiple of 16, >= dy), readable for m rounded up to 128 rows (the cache holds zeros for the
 * padded control points and cells, so the padding's values only have to be finite).  r: n values of the cell dtype.
 * workspace: mvf_wide_workspace_bytes(n, m).  Deterministic (fixed summation orders). */
size_t mvf_wide_workspace_bytes(int64_t n, int64_t m);
int mvf_rhs_cached(const void* ublk, const void* P, const void* Yd, int64_t n, int64_t m, int dy, int64_t ldy, double* R,
                   int64_t ldr, void* workspace, size_t workspace_bytes, mvf_dtype dtype, void* stream);
int mvf_apply_cached(const void* ublk, int64_t n, int64_t m, const double* C, int64_t ldc, int dy, const void* Yd, int64_t ldy,
                     const void* P, void* Vd, void* r, double* stats, void* workspace, size_t workspace_bytes,
                     mvf_dtype dtype, void* stream);

/* ---- coefficient solve ------------------------------------------------------------------------------------------
 * Replaces: dynamo `lstsq_solver(lhs, rhs, "scipy")` as Spateo calls it (sparsevfc.py:110,194,250) =
 * scipy.linalg.lstsq = LAPACK gelsd: the MINIMUM-NORM solution with singular values below eps * s_max dropped
 * (in-tree analogue: `_pinv(SigmaInv)` spateo/alignment/methods/morpho_class.py:1287).  lhs = G + lambda_sigma2 * K is
 * symmetric, numerically positive SEMI-definite and, at Spateo's default lambda_ = 0.02, rank deficient.  Two entry
 * points; the host (vectorfield.py) uses the first when it certifies full numerical rank and the second otherwise:
 *
 * mvf_solve: (G + lambda_sigma2 K + jitter * mean(diag) * I) C = R by blocked right-looking Cholesky in float64 (f64
 *   MFMA trailing update) + forward/back substitution.  With jitter = 0 and a matrix of full numerical rank this IS
 *   the gelsd solution (nothing is truncated).  info[0] = 0 on success, or 1 + index of the first non-positive pivot.
 *   pivots (may be NULL): [0] = min_j L_jj^2, [1] = max_j L_jj^2 - min L_jj^2 is an upper bound of lambda_min, the
 *   host's rank certificate.  G, K: m x m float64 (read-only), R: m x nrhs, C out: m x nrhs, nrhs <= 8.
 *
 * mvf_solve_minnorm: C = sum_{|lambda_i| > rcond max|lambda|} q_i (q_i^T R) / lambda_i over the eigenpairs of
 *   G + lambda_sigma2 K (for a symmetric matrix exactly gelsd's SVD-truncated minimum-norm solution; rcond =
 *   DBL_EPSILON reproduces scipy's default).  Hand-written symmetric eigensolver: Cholesky of the matrix shifted by
 *   delta = shift * mean(diag) (0 < shift < 1; the shift only makes the factorisation exist and is subtracted from the
 *   eigenvalues again), then one-sided block Jacobi (f64 MFMA Gram and update tiles, 64 x 64 subproblems in LDS) on
 *   the factor's columns until a whole sweep applies no rotation.  info[0] != 0: the shift was too small for this
 *   matrix (the caller retries with a larger one); C is then not written.  einfo (12 float64, device): [0] = sweeps
 *   (x.5 if max_sweeps was hit first), [1] = kept rank, [2] = max|lambda|, [3] = min kept |lambda|, [4] = delta,
 *   [5] = min lambda.  NOT asynchronous: it synchronises `stream` once per sweep to read the rotation counter.
 *   reuse != 0: the decomposition of the previous call on this workspace (same matrix) is applied to another R (a
 *   wide Y is solved in groups of <= 8 columns); asynchronous, einfo must hold 12 float64 ([6..11] = this call's).
 *   basis (may be NULL; mvf_solve_minnorm_basis_bytes, caller-owned): on exit the orthonormal eigenvectors (row i =
 *   eigenvector i).  warm != 0: on entry it holds the eigenvectors of a NEARBY matrix (the previous EM iteration's):
 *   the matrix is first transformed to that basis (f64 MFMA GEMMs), where its well-determined part is already
 *   diagonal - same result, about half the sweeps (measured 27 -> 13 at m = 3000, 18 -> 4 at m = 500; the eigenvectors
 *   of eigenvalues near the rounding level of the matrix are re-resolved against each factorisation's own rounding
 *   noise, which is what the remaining sweeps do). */
size_t mvf_solve_workspace_bytes(int64_t m, int nrhs);
int mvf_solve(const double* G, const double* K, double lambda_sigma2, double jitter, const double* R, int64_t m,
              int nrhs, double* C, int* info, double* pivots, void* workspace, size_t workspace_bytes, void* stream);
size_t mvf_solve_minnorm_workspace_bytes(int64_t m, int nrhs);
int mvf_solve_minnorm(const double* G, const double* K, double lambda_sigma2, double shift, double rcond,
                      const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo, int max_sweeps,
                      int reuse, double* basis, int warm, void* workspace, size_t workspace_bytes, void* stream);
size_t mvf_solve_minnorm_basis_bytes(int64_t m);

/* mvf_solve_minnorm_lr: the same truncated minimum-norm solve (same reference semantics: scipy.linalg.lstsq = gelsd,
 * spateo/tdr/morphometrics/morphofield/sparsevfc.py:110,194,250; `_pinv`, spateo/alignment/methods/morpho_class.py:1287)
 * through a RANK-REVEALING factor: greedy diagonally pivoted Cholesky  A = L L^T + E  (L m x r), stopped when every
 * remaining diagonal entry is <= tolf * DBL_EPSILON * lambda_max (lambda_max from 8 power-iteration steps; trace(E)
 * bounds ||E|| and sits at the rounding level of A), then one-sided block Jacobi on the r columns of L only, then
 * C = sum over sigma_i^2 > rcond max sigma^2 of u_i (u_i^T R) / sigma_i^2.  In the EM's steady state r ~ 0.3 m at
 * m = 3000 and the graded pivoted factor halves the sweep count; no shift, no retry ladder: info[0] != 0 only for
 * non-finite input.  einfo (12 float64, device): [0] = sweeps (x.5 if max_sweeps was hit first), [1] = kept rank,
 * [2] = max lambda, [3] = min kept lambda, [4] = 0, [5] = min lambda of the factor, [6] = r (columns of L).
 * rank_hint (0 = none): einfo[6] of the previous call ON THIS WORKSPACE for a nearby matrix (the previous EM iteration):
 * the factorisation then first follows the pivot order that call left in the workspace, 64 columns per three launches
 * (64 x 64 Cholesky of the gathered block, forward substitution of the gathered rows, trailing update) instead of one
 * launch per pivot, accepting a pivot only while it exceeds max(2^-10 x the largest remaining diagonal entry, the
 * stopping tolerance); the first rejected column ends the use of the hint and the greedy steps finish.  The result does
 * not depend on the hint being good: a stale or foreign order is rejected column by column, a workspace without a
 * finished order ignores the hint.  reuse != 0: apply the decomposition of the previous call on this workspace to another
 * R (einfo[7..11] = this call's).  NOT asynchronous (status reads during the factorisation, one per Jacobi sweep). */
size_t mvf_solve_minnorm_lr_workspace_bytes(int64_t m, int nrhs);
int mvf_solve_minnorm_lr(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond,
                         const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo, int max_sweeps,
                         int reuse, int rank_hint, void* workspace, size_t workspace_bytes, void* stream);

/* mvf_solve_minnorm_lrd: the same truncated minimum-norm solve (same reference call, same eps * lambda_max cut-off, same
 * arguments, same pivot-order / rank_hint / reuse behaviour and the same einfo layout as mvf_solve_minnorm_lr) WITHOUT the
 * eigendecomposition of the whole factor: after the pivoted Cholesky has dropped everything below tolf * eps * lambda_max,
 * the eigenvalues gelsd truncates are the few smallest ones of S2 = L^T L (r x r) and lie within 1 / tolf of the cut.
 * S2 = Rc Rc^T (Cholesky; the inverse factor rides along as extra rows), block inverse iteration on 256 vectors started on
 * the smallest pivots (three applications of S2^-1, Cholesky-QR in between), Rayleigh-Ritz on the 256 x 256 projection (the
 * Jacobi kernels; einfo[0] = ITS sweeps), W = Ritz vectors with theta <= rcond * lambda_max, Pc = I - W^T W, and
 *     C = L Pc S2^-1 Pc S2^-1 Pc L^T R
 * (the projections between the inverse applications keep the amplified rounding error of the dropped directions out).
 * lambda_max (einfo[2]) is the Rayleigh quotient of 12 power-iteration steps (relative error ~1e-8), einfo[3] the smallest
 * Ritz value above the cut inside the block, einfo[7] the block size used (256; 128 when the
 * previous call on this workspace - rank_hint > 0 - deflated at most 72 directions, repeated with 256 if it then finds more
 * than 80; 64 for factors of 128 .. 511 columns, accepted while at most 40 Ritz values lie below the cut; 0 when the Jacobi
 * path answered).  When the factor has fewer than 128 columns, when more than 224 (block 256) Ritz values
 * fall below the cut, or a factorisation meets a non-positive pivot, the call continues on mvf_solve_minnorm_lr's Jacobi
 * path and returns its result.  Measured at m = 3000 in the EM's steady state (r = 869): 6.7 ms against 22.9 ms, the field
 * within 1e-6 of the Jacobi path's on the same factor.  The workspace is larger (the r x r scratch); mvf_pinv_diag does not
 * accept a workspace left by this call (it needs every eigenpair): use mvf_solve_minnorm_lr for that.  No reference
 * interface changes: this is how `lstsq_solver(lhs, rhs, "scipy")` is evaluated.
 * Direct form (ABI 5; m <= 640, rank_hint == m, i.e. the previous call on this workspace kept ALL m columns - what m = 500
 * control points give): the pivoted factorisation of the next matrix in that order is an ordinary Cholesky of the permuted
 * matrix, A_perm = Rc Rc^T, and the blocked factorisation with the identity riding along yields E = Rc^-T in the same pass;
 * block inverse iteration on 64 vectors (continued from the previous call's block when that call had this form), Rayleigh-Ritz
 * through the factor (H = (Z Rc)(Z Rc)^T, one 64 x 64 Jacobi tile diagonalised in one launch), and
 *     C = Pi^T Pc E E^T Pc Pi R
 * with the inverse applied as its two triangular factors (the product E E^T would lose the factor's grading: 3 - 4 digits of
 * the field).  Accepted only if every pivot clears tolf * eps * lambda_max, lambda_max has converged and at most 40 Ritz values
 * lie below the cut; anything else re-runs the call in the factor form above.  The workspace then holds the unchanged pivot
 * order (mvf_lr_pivot_order works), einfo[0] = Rayleigh-Ritz launches (1), einfo[6] = m, einfo[7] = 64; reuse works.  Measured
 * at m = 500: 1.5 ms against 2.5 ms for the factor form and 3.6 ms for mvf_solve_minnorm (round 6: 0.9 ms).
 * einfo[8] (ABI 6) names the form that answered: 0 = the Jacobi path, 1 = the factor form, 2 = the direct form; einfo[9] = 0.
 *
 * mvf_solve_minnorm_lrd_async (ABI 6): the direct form WITHOUT a single host synchronisation - the synchronous entry point
 * reads the workspace state before it launches, the Rayleigh-Ritz counter and the acceptance numbers after (three round
 * trips of ~40 us in a 0.9 ms call, and the host cannot run ahead of the device across any of them).  The caller says what it
 * knows from the previous call's einfo ON THIS WORKSPACE: form_hint = 1: that call returned einfo[6] == m (all m columns
 * kept) through the factor form, 2: through the direct form (its block is continued); + 4 (that is 5 or 6): the matrix
 * moved since that call (the caller's sigma^2 changed by more than a few per cent): 13 warm power steps for lambda_max
 * instead of 5, as the synchronous entry adds them after reading the Rayleigh quotient.  The device verifies the state
 * (a stale or foreign workspace, a pending cool-down: not accepted) and takes the acceptance decision itself (every pivot
 * above the tolerance, lambda_max settled after the five warm power steps, the 64 x 64 Rayleigh-Ritz converged inside its one
 * launch, at most 40 directions below the cut): einfo[9] = 0: accepted - C, einfo[0..8] and the workspace exactly as the
 * synchronous direct form leaves them; einfo[9] = 1: NOT accepted - C is undefined and the caller must repeat the call
 * through mvf_solve_minnorm_lrd (the workspace carries the cool-down mark, so that call goes to the factor form at once).
 * Asynchronous on `stream`; einfo / info are read by the caller in its own device -> host copy.  128 <= m <= 640. */
size_t mvf_solve_minnorm_lrd_workspace_bytes(int64_t m, int nrhs);
int mvf_solve_minnorm_lrd(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond,
                          const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo, int max_sweeps,
                          int reuse, int rank_hint, void* workspace, size_t workspace_bytes, void* stream);
int mvf_solve_minnorm_lrd_async(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond,
                                const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo, int form_hint,
                                void* workspace, size_t workspace_bytes, void* stream);

/* The pivot order of the factorisation the last mvf_solve_minnorm_lr call left in `workspace` (same m): order_out (HOST,
 * room for m ints) receives the r pivots in the order they were taken, *r_out = r.  These are the control points that
 * carry the numerical rank of  U^T P U + lambda sigma^2 K;  the host's optional "pivot" Gram mode restricts the rest of
 * a fit to them.  pivots_out (HOST, m float64, may be NULL): the diagonal value each pivot had when it was taken (falling,
 * the last ones at the stopping tolerance); tol_out (HOST, may be NULL): that tolerance, tolf * eps * lambda_max.
 * Synchronises `stream`. */
int mvf_lr_pivot_order(const void* workspace, size_t workspace_bytes, int64_t m, int* order_out, double* pivots_out,
                       double* tol_out, int64_t* r_out, void* stream);

/* diag_out[n] = (U pinv(A) U^T)_nn, U = con_K(x, ctrl, beta), with the decomposition of A that the previous
 * mvf_solve_minnorm_lr (lowrank != 0) / mvf_solve_minnorm (lowrank == 0) call left in `workspace` (same m, same rcond
 * semantics: eigenvalues below rcond * max|lambda| dropped).  Replaces the last statement of
 * `Morpho_pairwise._update_nonrigid`, spateo/alignment/methods/morpho_class.py:1295-1297:
 * `SigmaDiag = sigma2 * einsum("ij->i", einsum("ij,ji->ij", U, dot(Sigma, U.T)))` (the caller multiplies by sigma2).
 * x4: n x 4 (dtype), ctrl4: m x 4 (dtype), diag_out: n float64.  Synchronises `stream` once on the lowrank path. */
int mvf_pinv_diag(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, double rcond, int lowrank,
                  double* diag_out, void* workspace, size_t workspace_bytes, mvf_dtype dtype, void* stream);

/* out[i] = a A[i] + b B[i] + c C[i], n float64 (B, C may be NULL; out may alias an input).  The compositions of
 * `Morpho_pairwise._update_nonrigid`'s SVI and guidance branches (spateo/alignment/methods/morpho_class.py:1269-1288):
 * `SigmaInv = step SigmaInv_new + (1 - step) SigmaInv_prev`, `SigmaInv += w U_I^T U_I`, `UPXB_term += w U_I^T (X_BI - R_AI)`. */
int mvf_lincomb3(double* out, double a, const double* A, double b, const double* B, double c, const double* C,
                 int64_t n, void* stream);

/* trace(C^T K C) -> out[0] (float64), the regulariser of the energy (App. A 5b). K: m x m, C: m x nrhs;
 * scratch >= m float64 (row partials, summed in row order). */
int mvf_quadform(const double* K, const double* C, int64_t m, int nrhs, double* out, double* scratch, void* stream);

/* Packed upper triangle (row-major, row i = columns i..m-1, m (m + 1) / 2 float64) <-> full symmetric m x m: the
 * multi-GPU host all-reduces the packed triangle of G (36 MB instead of 72 MB at m = 3000). */
int mvf_sym_pack(const double* G, int64_t m, double* tri, void* stream);
int mvf_sym_unpack(const double* tri, int64_t m, double* G, void* stream);

/* ---- the EM step's exchange: all-reduce of the sufficient statistics over RCCL (xGMI) -------------------------------
 * cells are block-sharded over the GPUs of one node, one process per GPU; per EM step every rank
 * contributes its partial  [tri(G) | R | sum P, sum P r, #(P > theta), ...]  (contiguous float64, caller-owned) and all
 * ranks continue with the sums.  The reference has no counterpart (single process, NumPy); the statement it distributes
 * is `lhs = UP.dot(U) ...; rhs = UP.dot(Y)` of SparseVFC (App. A 5c; call site
 * spateo/tdr/morphometrics/morphofield/sparsevfc.py:189-198) - sums over cells, hence an all-reduce of the partials.
 *   mvf_comm_unique_id : id_out = MVF_COMM_ID_BYTES host bytes (ncclGetUniqueId).  ONE rank calls it and hands the bytes to
 *                        the others out of band (the Python host broadcasts them with torch.distributed / its store).
 *   mvf_comm_create    : collective over all `nranks` ranks (ncclCommInitRank) on the CURRENT HIP device, which becomes the
 *                        communicator's device; nranks = 1 is valid (a single-rank communicator: how the one-GPU tests
 *                        execute this path on RCCL).  The handle owns nothing but the RCCL communicator.
 *   mvf_comm_destroy   : frees it (NULL is a no-op).
 *   mvf_comm_info      : nranks / rank / device of a handle AS RCCL REPORTS THEM (ncclCommCount / ncclCommUserRank /
 *                        ncclCommCuDevice; any pointer may be NULL) - what bench.py records as `rccl_ranks`.
 *   mvf_allreduce_stats: buf[0..count) (DEVICE float64, in place) <- elementwise SUM (MVF_RED_SUM) or MIN (MVF_RED_MIN: the
 *                        E-step's global min-non-zero rule) over the ranks; asynchronous on `stream` (ordered with the
 *                        kernels the caller launched there: pass a second stream + events to overlap it with compute, as
 *                        the host does for tri(G)).  The current device must be the communicator's.  Every rank must call
 *                        with the same count / op in the same order.  RCCL's sums are in a fixed ring order: every rank
 *                        receives bit-identical results, which the redundant coefficient solve relies on. */
typedef struct mvf_comm mvf_comm;
#define MVF_COMM_ID_BYTES 128
enum { MVF_RED_SUM = 0, MVF_RED_MIN = 1 };
int mvf_comm_unique_id(void* id_out);
int mvf_comm_create(mvf_comm** comm_out, int nranks, int rank, const void* id);
int mvf_comm_destroy(mvf_comm* comm);
int mvf_comm_info(const mvf_comm* comm, int* nranks, int* rank, int* device);
int mvf_allreduce_stats(mvf_comm* comm, double* buf, int64_t count, int op, void* stream);

/* ---- differential-geometry evaluators -------------------------------------------------------------------------
 * Replaces: dynamo `Jacobian_rkhs_gaussian` and `compute_{acceleration,curvature,curl,torsion,divergence}` =
 * in-tree twins spateo/tdr/morphometrics/morphofield_dg/GPVectorField.py:143-190 and :12-121 (wrappers
 * differential_geometry.py:42-341).  One fused kernel: per query point, one pass over the control points
 * accumulates v (3) and J (3x3) in float64, then derives the requested quantities in registers.
 * x4: n x 4 (dtype), ctrl4: m x 4 (dtype), C: m x 3 float64.  Outputs are float64, NULL if not requested:
 * v, curl, acc, curv, tors: n x 3;  jac: (3, 3, n) = J[f][i][n];  div, jdet: n.  `flags` = MVF_EVAL_* bits. */
int mvf_eval(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C, int flags,
             double* v, double* jac, double* div, double* curl, double* acc, double* curv, double* tors,
             double* jdet, mvf_dtype dtype, void* stream);

/* Same with an affine epilogue, for the Gaussian-process morphofield variant:
 *   v_out[f] = alpha[f] * (K @ C)[f] + (A q + b)[f],   J_out = jmul * J,   q = the query point as passed in x4 (unscaled);
 * every derived quantity (div, curl, acc, curvature, torsion, det) is computed from v_out / J_out.
 * `affine` is a HOST array of 16 doubles {alpha[3], jmul, A[9] row-major, b[3]} read at call time (NULL = identity);
 * alpha is per output component since ABI 6 (a per-axis `scale_fixed` of the GP variant's norm_dict).
 * Replaces: `_gp_velocity` spateo/tdr/morphometrics/morphofield/gaussian_process.py:102-127 and
 * `Jacobian_GP_gaussian_kernel` / `GPVectorField` morphofield_dg/GPVectorField.py:143-266 (norm_dict + rigid part). */
int mvf_eval_affine(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C,
                    const double* affine, int flags, double* v, double* jac, double* div, double* curl, double* acc,
                    double* curv, double* tors, double* jdet, mvf_dtype dtype, void* stream);

/* ---- trajectory integration (morphopath) ------------------------------------------------------------------------
 * Integrates dx/dt = v(x), v as in mvf_eval_affine, from the n start points x4 with classical RK4: n_out samples per
 * trajectory, `dt` apart, `substeps` RK4 steps between samples; traj (float64) = [n][n_out][3], traj[:, 0] = start.
 * One launch, one lane per trajectory, control points staged in LDS.  A negative dt integrates backwards.
 * Replaces the field evaluations inside dynamo `fate` as driven by `morphopath`
 * (spateo/tdr/morphometrics/morphofield/trajectory.py:61-109).  The integrator itself is this repo's (fixed-step RK4,
 * uniform time sampling): dynamo's adaptive RK45 + arc-length resampling lives outside the reference tree. */
int mvf_integrate(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C,
                  const double* affine, double dt, int substeps, int n_out, double* traj, mvf_dtype dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVF_H */
