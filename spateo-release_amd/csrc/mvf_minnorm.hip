// Minimum-norm coefficient solve with the reference's lstsq semantics:
//     C = sum over |lambda_i| > rcond * max|lambda|  of  q_i (q_i^T R) / lambda_i ,   (G + ls2 K) = Q diag(lambda) Q^T
//
// Reference: dynamo `lstsq_solver(lhs, rhs, "scipy")` = scipy.linalg.lstsq = LAPACK gelsd (minimum-norm solution,
// singular values below eps * s_max dropped), as Spateo calls it (spateo/tdr/morphometrics/morphofield/
// sparsevfc.py:110,194,250); in-tree analogue `_pinv(SigmaInv)` spateo/alignment/methods/morpho_class.py:1287.
// lhs is symmetric, so its singular values are |eigenvalues| and the SVD-truncated solution is the formula above.
//
// Three hand-written solvers for gfx950 (no rocSOLVER) share the kernels of this file:
//
// mvf_solve_minnorm_lrd (the host uses it from m = 640): step 1 of mvf_solve_minnorm_lr, then ONLY the invariant subspace
//   below the cut-off (block inverse iteration on the r x r matrix L^T L, Rayleigh-Ritz) and a deflated solve - see "deflated
//   truncated solve" below; falls back to mvf_solve_minnorm_lr's steps 2 - 3 when its block cannot hold that subspace.
//
// mvf_solve_minnorm_lr (the full decomposition of the same factor; what mvf_pinv_diag needs): rank-revealing.
//   1. A = L L^T + E, L m x r: greedy diagonally pivoted Cholesky (LAPACK pstrf's lazy scheme, one launch per pivot and one
//      MFMA trailing update per 64), stopped when every remaining diagonal entry is <= 0.25 eps lambda_max; from the second
//      call on a workspace it follows the previous call's pivot order in 64-column panels (three launches per 64 pivots).
//   2. one-sided block Jacobi on the r columns of L only;  3. as below with delta = 0.
//
// mvf_solve_minnorm (m < 640, where the factor keeps nearly every column and a warm start pays): full width.
//   1. A + delta I = L L^T      blocked Cholesky of mvf_solve.hip, delta = shift * mean(diag) > 0 only makes the
//                               factorisation exist (A is numerically semi-definite); it is subtracted again below.
//   2. one-sided block Jacobi on the COLUMNS of L (Veselic-Hari: orthogonalising L's columns diagonalises L^T L, one
//      LR step ahead of L L^T, and one-sided rotations give the small singular values to high RELATIVE accuracy).
//      Y = L^T row-major (row j = column j of L).  Columns are grouped in blocks of 32; a round-robin tournament pairs
//      the blocks; per round and pair:  S = Y_pair Y_pair^T (64 x 64, f64 MFMA, K split over workgroups, partials
//      summed in a fixed order) -> one pass of two-sided Jacobi rotations on S in LDS (relative threshold; the pairs
//      inside the blocks in the sweep's first round, the pairs between the two blocks in the others) gives an
//      orthogonal J -> Y_pair <- J^T Y_pair (f64 MFMA).  Sweeps repeat until a whole sweep applies no rotation.
//   3. rows of the final Y are x_i = sigma_i w_i with A + delta I = W diag(sigma^2) W^T, so lambda_i = sigma_i^2 -
//      delta and   C = Y^T ( g .* (Y R) ),   g_i = [|lambda_i| > rcond max|lambda|] / (sigma_i^2 lambda_i).
// Every transformation applied to Y is orthogonal to rounding, so Y^T Y == L L^T to rounding whatever the rotation
// choices were: the result is the truncated solve of a matrix within O(eps ||A||) of A, like gelsd's.
// mvf_pinv_diag evaluates diag(U pinv(A) U^T) from the Y either of the two Jacobi solvers left behind.
#include <cstring>
#include "mvf_common.h"
#include "mvf_solve.h"
#include "mvf_chol_dev.h"

namespace mvf {

// ---- small device -> host status reads through pinned staging -------------------------------------------------------------
// A status read into pageable memory costs a round trip PER COPY (tools/readback_probe.hip on this part: kernel + 2 pageable
// hipMemcpyAsync + synchronise 40 us, the same through pinned memory 22 us, kernel + synchronise alone 18.5 us); the factor form
// at M = 3000 makes about ten such reads per solve.  rb_copy stages a copy in a per-thread pinned page, rb_sync synchronises
// the stream once and hands the bytes to their destinations.  (The page is never freed: 8 KB per OS thread that ever ran a
// solve; freeing it from a thread-exit destructor would race the runtime's own shutdown.)
namespace {
struct Readback {
    char* pin = nullptr;
    bool tried = false;
    size_t off = 0;
    int np = 0;
    struct { void* dst; size_t off, n; } pend[16];
};
thread_local Readback g_rb;

hipError_t rb_copy(hipStream_t st, void* dst, const void* src, size_t n) {
    Readback& rb = g_rb;
    if (!rb.tried) {
        rb.tried = true;
        if (hipHostMalloc((void**)&rb.pin, 8192, hipHostMallocPortable) != hipSuccess) {
            rb.pin = nullptr;
            (void)hipGetLastError();
        }
    }
    if (!rb.pin || rb.np == 16 || rb.off + n > 8192) return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st);
    const hipError_t e = hipMemcpyAsync(rb.pin + rb.off, src, n, hipMemcpyDeviceToHost, st);
    rb.pend[rb.np].dst = dst, rb.pend[rb.np].off = rb.off, rb.pend[rb.np].n = n;
    ++rb.np;
    rb.off += (n + 63) & ~(size_t)63;
    return e;
}

hipError_t rb_sync(hipStream_t st) {
    Readback& rb = g_rb;
    const hipError_t e = hipStreamSynchronize(st);
    for (int i = 0; i < rb.np; ++i) memcpy(rb.pend[i].dst, rb.pin + rb.pend[i].off, rb.pend[i].n);
    rb.np = 0;
    rb.off = 0;
    return e;
}
}  // namespace

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int JB = 32;        // columns of L per block
constexpr int JP = 2 * JB;    // a pair of blocks = the 64 x 64 subproblem
constexpr int LDR = JP + 2;   // LDS stride of row-major MFMA operand tiles read as [row = lane&15][k = lane>>4]
constexpr int LDK = JP + 16;  // LDS stride of k-major MFMA operand tiles read as [k = lane>>4][col = lane&15]

// round-robin tournament of n (even) players: round r in [0, n-1), pair k in [0, n/2)
__device__ __forceinline__ void rr_pair(int n, int r, int k, int& a, int& b) {
    if (k == 0) {
        a = n - 1;
        b = r;
    } else {
        a = (r + k) % (n - 1);
        b = (r - k + (n - 1)) % (n - 1);
    }
}

// Y (mp x mp row-major) = L^T with everything outside the leading m x m lower triangle zeroed: Y[j][i] = L[i][j]
__global__ __launch_bounds__(256) void jac_init_kernel(const double* __restrict__ W, int64_t m, int64_t mp,
                                                       double* __restrict__ Y, int* __restrict__ zero_a = nullptr, int na = 0,
                                                       unsigned int* __restrict__ zero_b = nullptr, int nzb = 0) {
    __shared__ double t[64][65];
    const int bi = blockIdx.y, bj = blockIdx.x;  // tile of L: rows bi*64.., columns bj*64..
    if (bi == 0 && bj == 0) {  // the caller's Jacobi bookkeeping (stamps, rotation counters): saves two memset launches
        for (int e = threadIdx.x; e < na; e += 256) zero_a[e] = 0;
        for (int e = threadIdx.x; e < nzb; e += 256) zero_b[e] = 0u;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int64_t i = (int64_t)bi * 64 + r, j = (int64_t)bj * 64 + tx;
        t[r][tx] = (i < m && j < m && j <= i) ? W[i * mp + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t j = (int64_t)bj * 64 + r, i = (int64_t)bi * 64 + tx;
        Y[j * mp + i] = t[tx][r];
    }
}

__device__ __forceinline__ int64_t pair_row(int bp, int bq, int r) {
    return r < JB ? (int64_t)bp * JB + r : (int64_t)bq * JB + (r - JB);
}

// partial Gram tile of one block pair over a K range:  Spart[pair][split] = Y_pair[:, K range] Y_pair[:, K range]^T
// Convergence bookkeeping (device side): `mod[b]` = stamp of the last round in which block b was rotated, `clean[bp][bq]`
// = stamp of the last visit of that pair that found nothing to rotate.  A pair whose clean stamp is newer than both
// blocks' modification stamps is still orthogonal - its Gram tile would come out bit-identical - so its Gram, rotation
// and update work is skipped: the verification sweep that ends the iteration, and the converged pairs of the sweeps
// before it, cost next to nothing.
__device__ __forceinline__ bool pair_is_clean(const int* __restrict__ mod, const int* __restrict__ clean, int nb, int bp,
                                              int bq) {
    return clean[bp * nb + bq] > max(mod[bp], mod[bq]);
}

__global__ __launch_bounds__(256) void jac_gram_kernel(const double* __restrict__ Y, int64_t mp, int nb, int round,
                                                       int nsplit, int kchunks, const int* __restrict__ mod,
                                                       const int* __restrict__ clean, double* __restrict__ Spart) {
    const int pair = blockIdx.x, split = blockIdx.y;
    int bp, bq;
    rr_pair(nb, round, pair, bp, bq);
    if (pair_is_clean(mod, clean, nb, bp, bq)) return;
    __shared__ double sy[JP * LDR];  // one 64 x 64 tile (33 KB); the next tile waits in registers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nk = (int)(mp / 64);
    const int k0 = split * kchunks, k1 = min(nk, k0 + kchunks);
    // loader: wave w reads rows w, w + 4, ... of the 64-row pair tile, lane = column (512 contiguous bytes per row)
    const double* rowp[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) rowp[q] = Y + pair_row(bp, bq, wave + 4 * q) * mp + lane;
    double v[16];
    if (k0 < k1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = rowp[q][(int64_t)k0 * 64];
#pragma unroll
        for (int q = 0; q < 16; ++q) sy[(wave + 4 * q) * LDR + lane] = v[q];
    }
    __syncthreads();
    for (int kc = k0; kc < k1; ++kc) {
        const bool more = kc + 1 < k1;
        if (more) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = rowp[q][(int64_t)(kc + 1) * 64];
        }
        const double* s = sy;
#pragma unroll 4
        for (int kk = 0; kk < 64; kk += 4) {
            double fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                fa[a] = s[(wr + a * 16 + li) * LDR + kk + lk];  // A[i][k] = Yt[wr + i][k]
                fb[a] = s[(wc + a * 16 + li) * LDR + kk + lk];  // B[k][j] = Yt[wc + j][k]
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
#pragma unroll
            for (int q = 0; q < 16; ++q) sy[(wave + 4 * q) * LDR + lane] = v[q];
        }
        __syncthreads();
    }
    double* out = Spart + ((int64_t)pair * nsplit + split) * (JP * JP);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + a * 16 + lk + 4 * r;
                const int col = wc + b * 16 + li;
                out[row * JP + col] = acc[a][b][r];
            }
}

// Reciprocal / reciprocal square root from the hardware estimate (v_rcp_f64 / v_rsq_f64, ~2^-23) plus Newton steps:
// the rotation ANGLE only steers convergence, so one step (~2^-45) is plenty for it; orthogonality needs c^2 + s^2 = 1
// to rounding, which the two-step rsqrt of (1 + t^2) delivers.  (The IEEE sqrt / divide sequences the compiler emits are
// ~20 dependent instructions each, seven of them per rotation: they were the longest phase of this latency-bound kernel.)
__device__ __forceinline__ double rcp_nr(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    return y * fma(-x, y, 2.0);
}
__device__ __forceinline__ double rsq_nr(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    return y * fma(-0.5 * x * y, y, 1.5);
}
__device__ __forceinline__ double rsq_nr2(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y * fma(-0.5 * x * y, y, 1.5);
}

// Pair l of inner round r.  FULL: round-robin over all 64 columns (63 rounds: every pair, also the ones inside a
// block).  Cross-only: 32 rounds that pair column l of the first block with column (l + r) mod 32 of the second - the
// pairs BETWEEN the two blocks.  A sweep runs FULL in its first tournament round (each block is in exactly one pair
// there, so the pairs inside every block are visited once) and cross-only in the others: every column pair exactly
// once per sweep, the classical cyclic ordering, at half the inner rounds.
template <bool FULL>
__device__ __forceinline__ void inner_pair(int r, int l, int& p, int& q) {
    if (FULL) {
        rr_pair(JP, r, l, p, q);
    } else {
        p = l;
        q = JB + ((l + r) & (JB - 1));
    }
}

// One cyclic two-sided Jacobi sweep on the 64 x 64 Gram tile of a block pair, in LDS.  Rotation (p, q) is applied when
// |s_pq| > tol sqrt(s_pp s_qq) (the one-sided criterion: the two columns are not yet orthogonal relative to their
// norms).  1024 threads = 16 waves, four per SIMD: the kernel is a chain of short LDS round trips and dependent f64
// operations, so it needs several waves per SIMD to hide their latency (the first version ran one wave per SIMD with
// all of it exposed: 86 us per call against 41 us for the same work here, measured at M = 3000).  Thread (l = tid & 31, k = tid >> 5).
// Phase A: the 32 lanes of the first half-wave compute the round's 32 rotations and publish (c, s) in LDS.  Phase B:
// thread (l, k) updates the 2 x 2 block S_kl <- Rot_k^T S_kl Rot_l and the column pair l of J for rows k and k + 32.
// Blocks partition S, so phase B is in place; two barriers per round.  Within a half-wave the 32 lanes touch 32 distinct
// columns of one row: conflict-free without padding.
constexpr int EIG_THREADS = 1024;
// Stopping criterion of the RAYLEIGH-RITZ problems of the deflated solves (the projected matrix H = (Z Rc)(Z Rc)^T of the block
// of smallest Ritz values; NOT of the full eigendecompositions, which keep sqrt(m) eps): |h_pq| <= RR_TOL sqrt(h_pp h_qq).
#ifndef MVF_RR_TOL
#define MVF_RR_TOL 1.4551915228366852e-11 /* 2^-36 */
#endif
constexpr double RR_TOL = MVF_RR_TOL;
// inner_sweeps > 1 (FULL only; used when the pair IS the whole matrix, the 64 x 64 Rayleigh-Ritz problem of the small
// deflated solve): the sweep is repeated on the Gram tile in LDS until one applies no rotation (at most inner_sweeps times)
// and J accumulates all of them - a complete two-sided Jacobi diagonalisation in ONE launch instead of eight rounds of
// Gram / sweep / update / status read (well-conditioned tile: the one-sided re-orthogonalisation from the factor is not
// needed for accuracy there).  rot_total then receives the rotation count of the LAST sweep (0 = converged).
template <bool FULL>
__global__ __launch_bounds__(EIG_THREADS) void jac_eig_kernel(const double* __restrict__ Spart, int nsplit, double tol,
                                                              int nb, int round, int stamp, int* __restrict__ mod,
                                                              int* __restrict__ clean, double* __restrict__ Jbuf,
                                                              int* __restrict__ flags,
                                                              unsigned int* __restrict__ rot_total, int inner_sweeps = 1,
                                                              const int* __restrict__ warm_flag = nullptr,
                                                              double* __restrict__ Jkeep = nullptr) {
    __shared__ double S[JP][JP];
    __shared__ double Jm[JP][JP];
    __shared__ double cs[2][JB][2];
    __shared__ int sweep_rot;
    __shared__ int round_act[64], round_list[64], n_act;
    const int pair = blockIdx.x, tid = threadIdx.x;
    int bp, bq;
    rr_pair(nb, round, pair, bp, bq);
    if (pair_is_clean(mod, clean, nb, bp, bq)) {  // uniform over the workgroup; nobody writes these stamps this round
        if (tid == 0) flags[pair] = 0;
        return;
    }
#ifdef MVF_EIG_CLOCKS
    const unsigned long long c0 = wall_clock64();
#endif
    const double* sp = Spart + (int64_t)pair * nsplit * (JP * JP);
    // sum of the K-split partial tiles in split order; the loads of 8 splits x 4 elements are in flight together (the
    // partials were written by other compute units a moment ago: every dependent load is a trip beyond this XCD's L2 -
    // the plain loop spent 25 us here at 24 splits, as long as the 32 rotation rounds)
    {
        double acc4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int q0 = 0; q0 < nsplit; q0 += 8) {
            double v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    v[u][w] = (q0 + u < nsplit) ? sp[(int64_t)(q0 + u) * (JP * JP) + tid + w * EIG_THREADS] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) acc4[w] += v[u][w];
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int e = tid + w * EIG_THREADS;
            S[e >> 6][e & 63] = acc4[w];
            Jm[e >> 6][e & 63] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    // Warm start (inner-sweep form only; warm_flag[0] != 0): Jkeep holds the total rotation of the previous call's tile - the
    // 64 x 64 Rayleigh-Ritz problem of the previous EM iteration, a nearby matrix in the same basis - so S <- Jk^T S Jk is
    // already nearly diagonal and two or three sweeps finish what eight start from the identity (0.5 ms -> 0.2 ms).  The
    // products run in place: every thread forms its four entries in registers, a barrier, then the write.
    if (FULL && inner_sweeps > 1 && warm_flag != nullptr && Jkeep != nullptr && warm_flag[0] != 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int e = tid + w * EIG_THREADS;
            Jm[e >> 6][e & 63] = Jkeep[e];
        }
        __syncthreads();
        double t4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {  // T = S Jk
            const int e = tid + w * EIG_THREADS, i = e >> 6, j = e & 63;
            double a = 0.0;
            for (int kk = 0; kk < JP; ++kk) a = fma(S[i][kk], Jm[kk][j], a);
            t4[w] = a;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int e = tid + w * EIG_THREADS;
            S[e >> 6][e & 63] = t4[w];
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {  // S' = Jk^T T
            const int e = tid + w * EIG_THREADS, i = e >> 6, j = e & 63;
            double a = 0.0;
            for (int kk = 0; kk < JP; ++kk) a = fma(Jm[kk][i], S[kk][j], a);
            t4[w] = a;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int e = tid + w * EIG_THREADS;
            S[e >> 6][e & 63] = t4[w];
        }
        __syncthreads();
    }
    const int l = tid & 31, k = tid >> 5;
    const double tol2 = tol * tol;
    int nrot = 0;
    constexpr int NR = FULL ? JP - 1 : JB;
    // rotation of pair l of inner round r from the current S (the 32 lanes of the first half-wave)
    auto rotation = [&](int r, double (*csb)[2]) {
        int p, q;
        inner_pair<FULL>(r, l, p, q);
        double c = 1.0, sn = 0.0;
        const double app = S[p][p], aqq = S[q][q], apq = S[p][q];
        const bool act = apq != 0.0 && apq * apq > tol2 * fabs(app * aqq);
        if (act) {
            const double zeta = (aqq - app) * rcp_nr(2.0 * apq);
            const double z2 = fma(zeta, zeta, 1.0);
            const double t = copysign(rcp_nr(fabs(zeta) + z2 * rsq_nr(z2)), zeta);
            c = rsq_nr2(fma(t, t, 1.0));
            sn = c * t;
            if (!(fabs(sn) <= 1.0)) {  // overflow / NaN in the estimate chain (|zeta| astronomically large): no rotation
                c = 1.0;
                sn = 0.0;
            }
        }
        csb[l][0] = c;
        csb[l][1] = sn;
        nrot += __popcll(__ballot(act));  // the 32 lanes of this half-wave = the round's 32 pairs
    };
#ifdef MVF_EIG_CLOCKS
    const unsigned long long c1 = wall_clock64();
#endif
    int nrot_any = 0;
    // Which inner rounds have anything to do?  Every pair of the sweep is tested ONCE, in parallel, on the tile as the sweep
    // finds it (1024 threads: one or two pairs each); rounds without an active pair are not run at all (they used to cost
    // two barriers and a full update pass of S and J each: the later sweeps of an iteration - and most of a warm-started
    // one - are almost empty), and a sweep that finds no active pair anywhere ends the iteration without running: the
    // verification sweep costs one pass of tests instead of 63 rounds.  (A pair that only becomes active through the
    // rotations of this sweep is met by the next sweep's tests - threshold Jacobi; the stopping criterion is the same
    // test on every pair, now on one and the same tile.)
    auto pair_active = [&](int r, int ll) -> bool {
        int p, q;
        inner_pair<FULL>(r, ll, p, q);
        const double app = S[p][p], aqq = S[q][q], apq = S[p][q];
        return apq != 0.0 && apq * apq > tol2 * fabs(app * aqq);
    };
#pragma unroll 1
    for (int sw = 0; sw < (FULL ? inner_sweeps : 1); ++sw) {
    nrot = 0;
    if (tid < 64) round_act[tid] = 0;
    __syncthreads();
    {
        if (k < NR && pair_active(k, l)) round_act[k] = 1;  // (benign race: every writer stores 1)
        if (FULL && k + 32 < NR && pair_active(k + 32, l)) round_act[k + 32] = 1;
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: compact the active rounds, in order
        const bool a = tid < NR && round_act[tid] != 0;
        const unsigned long long b = __ballot(a);
        if (a) round_list[__popcll(b & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) n_act = __popcll(b);
    }
    __syncthreads();
    const int nact = n_act;
    if (nact > 0) {
    if (k == 0) rotation(round_list[0], cs[0]);
    __syncthreads();
    // Per round: (1) every thread rotates its 2 x 2 block of S; barrier; (2) the first half-wave computes the NEXT round's
    // rotations from the updated S while all threads rotate their column pair of J (J is not needed for the rotations:
    // the dependent f64 chain of (2) hides behind the LDS traffic of the J update); barrier.
#pragma unroll 1
    for (int ri = 0; ri < nact; ++ri) {
        const int r = round_list[ri];
        int p, q;
        inner_pair<FULL>(r, l, p, q);
        const double(*csr)[2] = cs[ri & 1];
        const double cl = csr[l][0], sl = csr[l][1], ck = csr[k][0], sk = csr[k][1];
        {
            int pk, qk;
            inner_pair<FULL>(r, k, pk, qk);
#ifdef MVF_EIG_NO_S  // measurement arm only (wrong results): the S update switched off
            if (tol < 0.0)
#endif
            {
            const double m00 = S[pk][p], m01 = S[pk][q], m10 = S[qk][p], m11 = S[qk][q];
            const double r00 = fma(ck, m00, -(sk * m10)), r01 = fma(ck, m01, -(sk * m11));
            const double r10 = fma(sk, m00, ck * m10), r11 = fma(sk, m01, ck * m11);
            double n00 = fma(cl, r00, -(sl * r01)), n01 = fma(sl, r00, cl * r01);
            double n10 = fma(cl, r10, -(sl * r11)), n11 = fma(sl, r10, cl * r11);
            if (k == l && sl != 0.0) {
                n01 = 0.0;
                n10 = 0.0;
            }
            S[pk][p] = n00;
            S[pk][q] = n01;
            S[qk][p] = n10;
            S[qk][q] = n11;
            }
        }
        __syncthreads();
        if (k == 0 && ri + 1 < nact) rotation(round_list[ri + 1], cs[(ri + 1) & 1]);
#ifdef MVF_EIG_NO_J  // measurement arm only (wrong results): the J accumulation switched off
        if (tol < 0.0)
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = k + 32 * j;
            const double a = Jm[i][p], b = Jm[i][q];
            Jm[i][p] = fma(cl, a, -(sl * b));
            Jm[i][q] = fma(sl, a, cl * b);
        }
        __syncthreads();
    }
    }
    if (inner_sweeps > 1) {  // uniform: did this sweep rotate anything?  (nrot lives in the first half-wave)
        if (tid == 0) {
            sweep_rot = nrot;
            if (sw < 12) {  // diagnostics (developer option lr_timing prints them): active rounds / rotations of this sweep
                rot_total[8 + 2 * sw] = (unsigned int)nact;
                rot_total[9 + 2 * sw] = (unsigned int)nrot;
            }
        }
        __syncthreads();
        const int sr = sweep_rot;
        nrot_any += sr;
        __syncthreads();  // everybody has read sweep_rot before the next sweep's thread 0 overwrites it
        if (sr == 0) break;
    } else {
        nrot_any = nrot;
    }
    }
#ifdef MVF_EIG_CLOCKS
    const unsigned long long c2 = wall_clock64();
#endif
    double* jo = Jbuf + (int64_t)pair * (JP * JP);
    for (int e = tid; e < JP * JP; e += EIG_THREADS) jo[e] = Jm[e >> 6][e & 63];
    if (FULL && inner_sweeps > 1 && Jkeep != nullptr)
        for (int e = tid; e < JP * JP; e += EIG_THREADS) Jkeep[e] = Jm[e >> 6][e & 63];
#ifdef MVF_EIG_CLOCKS
    if (tid == 0 && !FULL) {
        rot_total[4] = (unsigned int)(c1 - c0);
        rot_total[5] = (unsigned int)(c2 - c1);
        rot_total[6] = (unsigned int)(wall_clock64() - c2);
    }
#endif
    if (tid == 0) {
        // inner_sweeps = 1: nrot_any == nrot (this sweep's count).  inner_sweeps > 1: J carries every sweep (flag / stamps from
        // nrot_any), the host's convergence counter gets the LAST sweep's count
        const bool warmed = FULL && inner_sweeps > 1 && warm_flag != nullptr && Jkeep != nullptr && warm_flag[0] != 0;
        if (warmed) nrot_any += 1;  // J = Jk x (this call's rotations): the update must be applied even if no sweep rotated
        flags[pair] = nrot_any > 0;
        if (nrot_any > 0) {
            if (nrot > 0) atomicAdd(rot_total, (unsigned int)nrot);
            mod[bp] = stamp;  // each block is in exactly one pair per round: single writer
            mod[bq] = stamp;
        } else {
            clean[bp * nb + bq] = stamp;
        }
    }
}

// Y_pair[:, 64-column chunk] <- J^T Y_pair[:, chunk]   (skipped when the pair's sweep applied no rotation).
// J^T is the LDS-staged A operand; the B operand (the Y tile, k-major) is read straight from global memory: for one
// k-step the 16 lanes of a quarter-wave read 128 contiguous bytes of one row.
__global__ __launch_bounds__(256) void jac_update_kernel(double* __restrict__ Y, int64_t mp, int nb, int round,
                                                         const double* __restrict__ Jbuf,
                                                         const int* __restrict__ flags) {
    const int pair = blockIdx.x, chunk = blockIdx.y;
    if (!flags[pair]) return;
    int bp, bq;
    rr_pair(nb, round, pair, bp, bq);
    __shared__ double sj[JP * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* jp = Jbuf + (int64_t)pair * (JP * JP);
    {
        double vj[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) vj[q] = jp[(wave + 4 * q) * JP + lane];
#pragma unroll
        for (int q = 0; q < 16; ++q) sj[(wave + 4 * q) * LDK + lane] = vj[q];
    }
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    // B[k][n] = Yt[k][wc + 16 b + n]: this lane's 16 k-steps x 2 column blocks, all loads issued up front
    double fbv[16][2];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const double* yr = Y + pair_row(bp, bq, 4 * ks + lk) * mp + (int64_t)chunk * 64 + wc + li;
        fbv[ks][0] = yr[0];
        fbv[ks][1] = yr[16];
    }
    __syncthreads();
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        double fa[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = sj[(4 * ks + lk) * LDK + wr + a * 16 + li];  // A[i][k] = J[k][wr + i]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fbv[ks][b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();  // every wave holds its inputs in registers before any wave overwrites the tile in place
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + a * 16 + lk + 4 * r;
                const int col = wc + b * 16 + li;
                Y[pair_row(bp, bq, row) * mp + (int64_t)chunk * 64 + col] = acc[a][b][r];
            }
}

// sig2[i] = ||Y_i||^2 and T[i][d] = Y_i . R[:, d]   (one wave per row of Y)
__global__ __launch_bounds__(256) void jac_rowstat_kernel(const double* __restrict__ Y, int64_t nrows, int64_t mp,
                                                          int64_t m, const double* __restrict__ R, int nrhs,
                                                          double* __restrict__ sig2, double* __restrict__ T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    double s = 0.0, t[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) t[d] = 0.0;
    const double* y = Y + row * mp;
    for (int64_t n = lane; n < m; n += 64) {
        const double v = y[n];
        s = fma(v, v, s);
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (d < nrhs) t[d] = fma(v, R[n * nrhs + d], t[d]);
    }
    s = wave_sum(s);
#pragma unroll
    for (int d = 0; d < 8; ++d) t[d] = wave_sum(t[d]);
    if (lane == 0) {
        sig2[row] = s;
#pragma unroll
        for (int d = 0; d < 8; ++d) T[row * 8 + d] = t[d];
    }
}

// lambda_i = sig2_i - delta;  truncate at rcond * max|lambda|;  T[i] *= 1 / (sig2_i lambda_i);  diagnostics -> einfo
// einfo: [1] = kept rank, [2] = max|lambda|, [3] = min kept |lambda|, [4] = delta, [5] = min lambda (signed)
__global__ __launch_bounds__(256) void jac_scale_kernel(const double* __restrict__ sig2, int64_t mp,
                                                        const double* __restrict__ scal, double rcond,
                                                        double* __restrict__ T, double* __restrict__ einfo) {
    __shared__ double red[4];
    __shared__ double bc;
    const double delta = scal[1];
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < mp; i += 256)
        if (sig2[i] > 0.0) mx = fmax(mx, fabs(sig2[i] - delta));
    const double t = -block_min<256>(-mx, red);
    if (threadIdx.x == 0) bc = t;
    __syncthreads();
    const double lmax = bc, cut = rcond * lmax;
    double kept = 0.0, mink = INFINITY, minl = INFINITY;
    for (int64_t i = threadIdx.x; i < mp; i += 256) {
        const double s2 = sig2[i], lam = s2 - delta;
        const bool keep = s2 > 0.0 && fabs(lam) > cut;
        const double g = keep ? 1.0 / (s2 * lam) : 0.0;
#pragma unroll
        for (int d = 0; d < 8; ++d) T[i * 8 + d] *= g;
        if (keep) {
            kept += 1.0;
            mink = fmin(mink, fabs(lam));
        }
        if (s2 > 0.0) minl = fmin(minl, lam);
    }
    const double k1 = block_sum<256>(kept, red);
    const double k2 = block_min<256>(mink, red);
    const double k3 = block_min<256>(minl, red);
    if (threadIdx.x == 0) {
        einfo[1] = k1;
        einfo[2] = lmax;
        einfo[3] = k2;
        einfo[4] = delta;
        einfo[5] = k3;
    }
}

// part[split][n][d] = sum over this split's rows i of Y[i][n] T[i][d]   (64 columns n per workgroup)
__global__ __launch_bounds__(256) void jac_back_kernel(const double* __restrict__ Y, int64_t nrows, int64_t mp,
                                                       int64_t m, const double* __restrict__ T, int rows_per_split,
                                                       double* __restrict__ part) {
    __shared__ double red[4][64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + lane;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_split, i1 = min(nrows, i0 + rows_per_split);
    double acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.0;
    if (n < mp) {
        for (int64_t i = i0 + wave; i < i1; i += 4) {
            const double y = Y[i * mp + n];
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] = fma(y, T[i * 8 + d], acc[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) red[wave][lane][d] = acc[d];
    __syncthreads();
    if (wave == 0 && n < m) {
#pragma unroll
        for (int d = 0; d < 8; ++d)
            part[((int64_t)blockIdx.y * m + n) * 8 + d] = red[0][lane][d] + red[1][lane][d] + red[2][lane][d] + red[3][lane][d];
    }
}

__global__ __launch_bounds__(256) void jac_back_reduce_kernel(const double* __restrict__ part, int nsplit, int64_t m,
                                                              int nrhs, double* __restrict__ C) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * nrhs) return;
    const int64_t n = e / nrhs;
    const int d = (int)(e % nrhs);
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += part[((int64_t)q * m + n) * 8 + d];
    C[e] = s;
}

// ---- warm start: the eigenvector basis of the previous EM iteration's matrix pre-diagonalises this one's ---------
// A changes little between EM iterations (only P and sigma^2 move), so A' = Wt A Wt^T with the previous eigenvectors
// (rows of Wt) is nearly diagonal and the Jacobi sweeps on chol(A' + delta I) start in their quadratic phase.

// Afull (mp x mp) = G + ls2 K on the leading m x m, zero on the padding
__global__ __launch_bounds__(256) void assemble_kernel(const double* __restrict__ G, const double* __restrict__ K,
                                                       double ls2, int64_t m, int64_t mp, double* __restrict__ A,
                                                       double* __restrict__ zero32 = nullptr) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (zero32 != nullptr && i == 0 && j < 32) zero32[j] = 0.0;  // (a 256-byte scalar slot of the caller: saves its memset launch)
    if (j >= mp) return;
    A[i * mp + j] = (i < m && j < m) ? G[i * m + j] + ls2 * K[i * m + j] : 0.0;
}

// C (rows x cols, leading dimension ldc) = op(A) op(B) over K, row-major, f64 MFMA, one 64 x 64 output tile per workgroup
// (grid = cols / 64, rows / 64), K (a multiple of 32) in stages of 32 through LDS.
//   TA = false: A is rows x K (lda);   TA = true: A is given transposed, K x rows (lda) - C = A^T op(B)
//   TB = false: B is K x cols (ldb);   TB = true: B is cols x K (ldb)                   - C = op(A) B^T
constexpr int GK = 32;
constexpr int GLR = GK + 2;   // row-major stage stride: [64 rows][32 k]
constexpr int GLK = JP + 16;  // k-major stage stride:   [32 k][64 cols]
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                   int64_t ldb, double* __restrict__ C, int64_t ldc, int64_t K,
                                                   int symmetric = 0) {
    // symmetric != 0 (the caller vouches that op(A) op(B) is symmetric - S2 = Y Y^T, Minv = E E^T): only the tiles on and
    // below the diagonal are computed, each is stored together with its mirror image (the result is exactly symmetric)
    if (symmetric && blockIdx.x > blockIdx.y) return;
    __shared__ double sa[TA ? GK * GLK : 64 * GLR];
    __shared__ double sb[TB ? 64 * GLR : GK * GLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    // loaders: row-major stage = 64 rows x 32 k: thread -> row tid >> 2, 8 consecutive k;  k-major stage = 32 k x 64
    // columns: thread -> k row tid >> 3, 8 consecutive columns
    const int rr = tid >> 2, rc = (tid & 3) * 8;
    const int kr = tid >> 3, kc = (tid & 7) * 8;
    double va[8], vb[8];
    auto load_stage = [&](int64_t k0) {
        const double* ap = TA ? A + (k0 + kr) * lda + i0 + kc : A + (i0 + rr) * lda + k0 + rc;
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = ap[q];
        const double* bp = TB ? B + (j0 + rr) * ldb + k0 + rc : B + (k0 + kr) * ldb + j0 + kc;
#pragma unroll
        for (int q = 0; q < 8; ++q) vb[q] = bp[q];
    };
    load_stage(0);
    for (int64_t k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();  // the previous stage has been consumed
        if (TA) {
#pragma unroll
            for (int q = 0; q < 8; ++q) sa[kr * GLK + kc + q] = va[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) sa[rr * GLR + rc + q] = va[q];
        }
        if (TB) {
#pragma unroll
            for (int q = 0; q < 8; ++q) sb[rr * GLR + rc + q] = vb[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) sb[kr * GLK + kc + q] = vb[q];
        }
        __syncthreads();
        if (k0 + GK < K) load_stage(k0 + GK);  // the next stage's global loads fly during this stage's MFMAs (round 5 issued
                                               // them after it: one exposed memory round trip per 32 values of k)
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            double fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                fa[a] = TA ? sa[(kk + lk) * GLK + wr + a * 16 + li]   // A^T given: At[k][i]
                           : sa[(wr + a * 16 + li) * GLR + kk + lk];  // A[i][k]
                fb[a] = TB ? sb[(wc + a * 16 + li) * GLR + kk + lk]   // B^T: B[j][k]
                           : sb[(kk + lk) * GLK + wc + a * 16 + li];  // B[k][j]
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + a * 16 + lk + 4 * r;
                const int col = wc + b * 16 + li;
                C[(i0 + row) * ldc + j0 + col] = acc[a][b][r];
                if (symmetric && blockIdx.x != blockIdx.y) C[(j0 + col) * ldc + i0 + row] = acc[a][b][r];
            }
}

// The same product for SKINNY shapes (the 64 / 128 / 256-vector block of the deflated solve against an rp x rp operand:
// 8 - 56 output tiles of 64 x 64 on a 256-CU part, each with the whole K loop behind it - 32 us per launch at rp = 512, five
// of them in every small solve).  One 16 x 16 output tile per workgroup, the four waves split K and their partial tiles are
// summed through LDS in wave order (deterministic); MFMA operand fragments come straight from global memory (the operands
// are L2-resident and every tile is read once per workgroup), 16 k-steps of loads in flight per wave ahead of the MFMAs.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb, double* __restrict__ C,
                                                          int64_t ldc, int64_t K) {
    __shared__ double part[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * 16, j0 = (int64_t)blockIdx.x * 16;
    const int64_t kq = K / 4, kb = wave * kq;  // this wave's K range (K is a multiple of 64)
    const double* ap = TA ? A + (kb + lk) * lda + i0 + li : A + (i0 + li) * lda + kb + lk;
    const double* bp = TB ? B + (j0 + li) * ldb + kb + lk : B + (kb + lk) * ldb + j0 + li;
    const int64_t sa = TA ? 4 * lda : 4, sb = TB ? 4 : 4 * ldb;  // pointer step per k-step (4 values of k)
    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nst = (int)(kq / 4);  // k-steps of this wave (a multiple of 4)
    double fa[16], fb[16];
    const int first = min(16, nst);
#pragma unroll
    for (int s = 0; s < 16; ++s)
        if (s < first) {
            fa[s] = ap[s * sa];
            fb[s] = bp[s * sb];
        }
    for (int s0 = 0; s0 < nst; s0 += 16) {
        double na[16], nb[16];
        const int cnt = min(16, nst - s0), ncnt = min(16, nst - s0 - 16);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < ncnt) {
                na[s] = ap[(s0 + 16 + s) * sa];
                nb[s] = bp[(s0 + 16 + s) * sb];
            }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < cnt) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            fa[s] = na[s];
            fb[s] = nb[s];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][(lk + 4 * r) * 16 + li] = acc[r];
    __syncthreads();
    const double v = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    C[(i0 + (tid >> 4)) * ldc + j0 + (tid & 15)] = v;
}

template <bool TA, bool TB>
static void gemm(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                 int64_t rows, int64_t cols, int64_t K, int symmetric = 0) {
    if ((rows / 64) * (cols / 64) < 96)  // fewer 64 x 64 tiles than would keep the part busy: the skinny form
        hipLaunchKernelGGL((gemm_skinny_kernel<TA, TB>), dim3((unsigned)(cols / 16), (unsigned)(rows / 16)), dim3(256), 0, st, A,
                           lda, B, ldb, C, ldc, K);
    else
        hipLaunchKernelGGL((gemm_kernel<TA, TB>), dim3((unsigned)(cols / 64), (unsigned)(rows / 64)), dim3(256), 0, st, A, lda,
                           B, ldb, C, ldc, K, symmetric);
}

// Wt[i][:] = Y[i][:] / sigma_i  (row i = eigenvector i);  the zero rows of the padding become unit vectors
__global__ __launch_bounds__(256) void basis_extract_kernel(const double* __restrict__ Y, const double* __restrict__ sig2,
                                                            int64_t mp, double* __restrict__ Wt) {
    const int64_t i = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= mp) return;
    const double s2 = sig2[i];
    Wt[i * mp + n] = s2 > 0.0 ? Y[i * mp + n] * (1.0 / sqrt(s2)) : (n == i ? 1.0 : 0.0);
}


// ================= rank-revealing path: diagonally pivoted Cholesky + one-sided Jacobi on the r kept columns ===========
// A (numerically rank r << m: r ~ 0.3 m at m = 3000 in the EM's steady state) = L L^T + E with L m x r from the greedy
// (largest remaining diagonal) pivoted Cholesky, stopped when every remaining diagonal is <= tol = tolf eps lambda_max:
// trace(E) bounds ||E|| and sits at the rounding level of A itself, below the eps lambda_max cut-off the truncated solve
// applies anyway.  The Jacobi iteration then orthogonalises r columns instead of m - and the pivoted factor is graded
// (column norms fall with the pivots), which is the Veselic-Hari / Drmac preconditioner: 13 cold sweeps at m = 3000 where
// the unpivoted shifted factor needs 26 - and no shift delta, no retry ladder.
// Y (row j = column j of L, row length mp) is produced row by row: coalesced stores, and exactly the layout the Jacobi
// kernels want.
constexpr int PCHOL_MAGIC_V = 0x6d766c72;  // == PCHOL_MAGIC (declared with the factorisation kernels below)
struct PcholState {
    int done, r, panels, hint_broken;  // panels = hint panels accepted so far (the next panel's index)
    double tol, lmax_est, maxdiag, lmax_prev;  // lmax_prev: the Rayleigh quotient one power step earlier
    int panel_nvalid, magic, order_len, keep_len;  // magic / order_len: the workspace holds the pivot order of a finished call;
                                                    // keep_len: length of the kept dominant eigenvector (warm power iteration)
    int defl, defl_block, pad4, defl_nsel;      // defl = 1: the workspace holds the DEFLATED decomposition (mvf_solve_minnorm_lrd)
                                                // of block size defl_block; defl_nsel: directions its last call deflated
    int direct_skip, pad5, pad6, pad7;          // > 0: so many further calls skip the direct form (its last attempt failed)
};

// y = A x, one wave per row (the power iteration that estimates lambda_max for the stopping tolerance)
__global__ __launch_bounds__(256) void lr_symv_kernel(const double* __restrict__ A, int64_t mp,
                                                      const double* __restrict__ x, double* __restrict__ y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= mp) return;
    const double* a = A + row * mp;
    double s = 0.0;
    for (int64_t n = lane; n < mp; n += 64) s = fma(a[n], x[n], s);
    s = wave_sum(s);
    if (lane == 0) y[row] = s;
}

// first = 1: x = 1 / sqrt(m) on the live entries.  Else: est = x^T y (x has unit norm: the Rayleigh quotient, a lower
// bound of lambda_max that converges from below), x = y / ||y||.
// first = 2 (warm): x = `keep`, the dominant eigenvector the previous call on this workspace ended its power iteration with -
// taken only if the workspace really holds a finished call of that factor rank (the same test as the pivot-order hint), else
// the cold start.  keep_out (may be NULL, first = 0 only): the normalised iterate is also written there.
__global__ __launch_bounds__(256) void lr_power_kernel(double* __restrict__ x, const double* __restrict__ y, int64_t m,
                                                       int64_t mp, int first, PcholState* __restrict__ stt,
                                                       const double* __restrict__ keep = nullptr, int hint_len = 0,
                                                       double* __restrict__ keep_out = nullptr) {
    __shared__ double red[4];
    __shared__ double bc[2];
    if (first) {
        const bool warm = first == 2 && keep != nullptr && stt->magic == PCHOL_MAGIC_V && stt->order_len == hint_len &&
                          stt->keep_len == (int)m;
        const double v = 1.0 / sqrt((double)m);
        for (int64_t i = threadIdx.x; i < mp; i += 256) x[i] = i < m ? (warm ? keep[i] : v) : 0.0;
        if (threadIdx.x == 0) stt->lmax_est = stt->lmax_prev = 0.0;
        return;
    }
    double xy = 0.0, yy = 0.0;
    for (int64_t i = threadIdx.x; i < mp; i += 256) {
        xy = fma(x[i], y[i], xy);
        yy = fma(y[i], y[i], yy);
    }
    const double t0 = block_sum<256>(xy, red);
    const double t1 = block_sum<256>(yy, red);
    if (threadIdx.x == 0) {
        bc[0] = t0;
        bc[1] = t1;
    }
    __syncthreads();
    const double inv = bc[1] > 0.0 ? 1.0 / sqrt(bc[1]) : 0.0;
    for (int64_t i = threadIdx.x; i < mp; i += 256) {
        const double v = y[i] * inv;
        x[i] = v;
        if (keep_out) keep_out[i] = v;
    }
    if (threadIdx.x == 0) {
        stt->lmax_prev = stt->lmax_est;
        stt->lmax_est = bc[0];
        if (keep_out) stt->keep_len = (int)m;
    }
}

constexpr int PC_T = 128;  // threads (= columns of A) per workgroup of the pivot step
constexpr int PCHOL_MAGIC = PCHOL_MAGIC_V;
constexpr double PCHOL_THETA = 0.0009765625;  // 2^-10, see pchol_panel_factor_kernel

// dg[i] = A_ii (-inf on the padding: never a pivot), per-workgroup (max, argmax) partials, the tolerance, the state
__global__ __launch_bounds__(256) void pchol_init_kernel(const double* __restrict__ S, int64_t m, int64_t mp, double tolf,
                                                         double* __restrict__ dg, double* __restrict__ pm, int nwg,
                                                         PcholState* __restrict__ stt, int* __restrict__ info, int use_hint,
                                                         int hint_len) {
    __shared__ double red[4];
    __shared__ double bc[2];
    double mx = 0.0;
    bool finite = true;
    for (int64_t i = threadIdx.x; i < mp; i += 256) {
        const double d = i < m ? S[i * mp + i] : -INFINITY;
        dg[i] = d;
        if (i < m) {
            finite = finite && (fabs(d) <= 1.79e308);
            mx = fmax(mx, d);
        }
    }
    const double bad = block_sum<256>(finite ? 0.0 : 1.0, red);
    const double t = -block_min<256>(-mx, red);
    if (threadIdx.x == 0) {
        bc[0] = bad;
        bc[1] = t;
    }
    __syncthreads();
    // partials: workgroup w of the step kernel owns columns [w PC_T, (w + 1) PC_T)
    for (int w = threadIdx.x; w < nwg; w += 256) {
        double bv = -INFINITY;
        int bi = w * PC_T;
        for (int q = 0; q < PC_T; ++q) {
            const int64_t i = (int64_t)w * PC_T + q;
            const double d = (i < m) ? S[i * mp + i] : -INFINITY;
            if (d > bv) {
                bv = d;
                bi = (int)i;
            }
        }
        pm[2 * w] = bv;
        pm[2 * w + 1] = (double)bi;
    }
    if (threadIdx.x == 0) {
        const double est = stt->lmax_est;
        const bool ok = bc[0] == 0.0 && (fabs(est) <= 1.79e308);
        const double lmax = fmax(ok ? est : 0.0, bc[1]);
        stt->done = 0;
        stt->r = 0;
        stt->panels = 0;
        // the hint is usable only if this workspace really holds the order of a finished factorisation of that length
        stt->hint_broken = !(use_hint && stt->magic == PCHOL_MAGIC && stt->order_len == hint_len);
        // a workspace without a finished factorisation of that length (fresh, foreign, other m) carries no deflation count
        // either: the deflated solve's block choice must be a function of the inputs, not of uninitialised memory
        if (stt->hint_broken) {
            stt->defl_nsel = 0;
            stt->direct_skip = 0;
        }
        stt->magic = 0;
        stt->panel_nvalid = 0;
        stt->maxdiag = bc[1];
        stt->lmax_est = lmax;
        stt->tol = tolf * 2.220446049250313e-16 * lmax;
        info[0] = ok ? 0 : 1;
    }
}

// One pivot step (launch j): p = argmax of the remaining diagonal (from the per-workgroup partials the previous launch
// left; ties -> lowest index); row j of Y = (S[p, :] - sum over this block's earlier rows k of Y[k, :] Y[k, p]) / sqrt(pivot)
// (S carries the trailing updates of all earlier blocks: LAPACK pstrf's lazy scheme); dg -= y^2.  Every thread
// accumulates the pivot value itself (same scalars, same order: bit-identical across the grid), so there is no second
// pass and no communication between workgroups inside a step.  dg / pm are double-buffered between launches.
__global__ __launch_bounds__(PC_T) void pchol_step_kernel(const double* __restrict__ S, double* __restrict__ Y, int64_t mp,
                                                          int jb, int j, const double* __restrict__ dg_in,
                                                          double* __restrict__ dg_out, const double* __restrict__ pm_in,
                                                          double* __restrict__ pm_out, int nwg,
                                                          PcholState* __restrict__ stt, int* __restrict__ order,
                                                          double* __restrict__ piv) {
    if (stt->done) return;
    const int tid = threadIdx.x, lane = tid & 63;
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int q = lane; q < nwg; q += 64) {
        const double v = pm_in[2 * q];
        const int ix = (int)pm_in[2 * q + 1];
        if (v > bv || (v == bv && ix < bi)) {
            bv = v;
            bi = ix;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(bv, o, 64);
        const int ix = __shfl_xor(bi, o, 64);
        if (v > bv || (v == bv && ix < bi)) {
            bv = v;
            bi = ix;
        }
    }
    const double tol = stt->tol;
    if (!(bv > tol)) {
        if (blockIdx.x == 0 && tid == 0) stt->done = 1;
        return;
    }
    const int64_t p = bi;
    const int64_t i = (int64_t)blockIdx.x * PC_T + tid;
    const bool live = i < mp;
    const int64_t ic = live ? i : 0;
    double acc = S[p * mp + ic], accp = S[p * mp + p];
    const double di = dg_in[ic];
    const double* yk = Y + (int64_t)jb * mp;
    int k = jb;
    // (measured: a step costs ~7 us whatever its size - at m = 500 as at m = 3000, with 4- or 16-row load batches: it is
    // the dependent-launch latency of the stream, not this loop)
    for (; k + 4 <= j; k += 4, yk += 4 * mp) {
        const double p0 = yk[p], p1 = yk[mp + p], p2 = yk[2 * mp + p], p3 = yk[3 * mp + p];
        const double a0 = yk[ic], a1 = yk[mp + ic], a2 = yk[2 * mp + ic], a3 = yk[3 * mp + ic];
        acc = fma(-a0, p0, acc);
        accp = fma(-p0, p0, accp);
        acc = fma(-a1, p1, acc);
        accp = fma(-p1, p1, accp);
        acc = fma(-a2, p2, acc);
        accp = fma(-p2, p2, accp);
        acc = fma(-a3, p3, acc);
        accp = fma(-p3, p3, accp);
    }
    for (; k < j; ++k, yk += mp) {
        const double p0 = yk[p], a0 = yk[ic];
        acc = fma(-a0, p0, acc);
        accp = fma(-p0, p0, accp);
    }
    const bool used = !live || di == -INFINITY;
    // a pivot the fresh evaluation does not confirm (dg drifted): retire p with an all-zero row
    const bool badp = !(accp > 0.25 * tol);
    const double inv = badp ? 0.0 : 1.0 / sqrt(accp);
    const double y = (used || badp) ? 0.0 : acc * inv;
    const double dn = (used || i == p) ? -INFINITY : di - y * y;
    if (live) {
        Y[(int64_t)j * mp + i] = y;
        dg_out[i] = dn;
    }
    // this workgroup's (max, argmax) of the new diagonal for the next launch
    __shared__ double sv[PC_T / 64];
    __shared__ int si[PC_T / 64];
    double wv = live ? dn : -INFINITY;
    int wi = live ? (int)i : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(wv, o, 64);
        const int ix = __shfl_xor(wi, o, 64);
        if (v > wv || (v == wv && ix < wi)) {
            wv = v;
            wi = ix;
        }
    }
    if (lane == 0) {
        sv[tid >> 6] = wv;
        si[tid >> 6] = wi;
    }
    __syncthreads();
    if (tid == 0) {
        double v = sv[0];
        int ix = si[0];
#pragma unroll
        for (int w = 1; w < PC_T / 64; ++w)
            if (sv[w] > v || (sv[w] == v && si[w] < ix)) {
                v = sv[w];
                ix = si[w];
            }
        pm_out[2 * blockIdx.x] = v;
        pm_out[2 * blockIdx.x + 1] = (double)ix;
        if (blockIdx.x == 0) {
            order[j] = (int)p;
            piv[j] = accp;
            stt->r = j + 1;
        }
    }
}

// ---- pivot steps of a 32-row sub-block in ONE launch by ONE workgroup (mp <= 512 columns: thread i owns column i) ----------
// A pivot step launched on its own costs ~8 us whatever its size, so the cold factorisation of a fit's FIRST EM iteration (no pivot
// order to follow yet) was m dependent launches: 4.3 ms at m = 500, a quarter of a whole BASELINE config 2 call.  Cycle counters in
// a first single-workgroup version (same loop, one launch per 64 steps: 3.98 ms) showed that it is NOT launch latency: per step
// 3200 cycles of shuffle-based argmax, 230 cycles per earlier row of dependent global loads (the step re-reads the block's rows it
// needs: 9300 at row 40), 3900 of IEEE sqrt / divide, the row's store and a barrier that waits for it.  What this kernel does instead:
//   * the rows of the current 32-row sub-block live in LDS (128 KB) - a step reads them from there, this column's entry and the
//     pivot column's (a broadcast); the rows go to global memory as well, but nothing waits for those stores: the barriers of a
//     step order LDS traffic only;
//   * the other half of a 64-row block (32 older rows): this column's entries once per launch into registers, the pivot column's
//     through LDS from one load instruction of the first wave - issued together with the pivot's row of S, the one global round
//     trip of a step;
//   * argmax by DPP row operations + a ballot (value first, then the lowest lane that holds it) instead of 12 dependent
//     ds_bpermute per wave.
// The arithmetic per column is pchol_step_kernel's, statement by statement and in the same order (rows ascending): same pivots,
// same rows, same bits.
constexpr int PS_T = 512;    // threads = columns
constexpr int PS_SUB = 32;   // rows per launch / LDS-resident sub-block
constexpr size_t PS_LDS = ((size_t)PS_SUB * PS_T + PS_SUB + 16) * sizeof(double) + 16 * sizeof(int);

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(b & 0xffffffffLL), (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// max over the wave, returned in every lane (through lane 63 and a scalar broadcast)
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_f64<0xB1, 0xf>(v));   // quad_perm [1, 0, 3, 2]
    v = fmax(v, dpp_f64<0x4E, 0xf>(v));   // quad_perm [2, 3, 0, 1]
    v = fmax(v, dpp_f64<0x124, 0xf>(v));  // row_ror 4
    v = fmax(v, dpp_f64<0x128, 0xf>(v));  // row_ror 8: every lane holds its row's maximum
    v = fmax(v, dpp_f64<0x142, 0xa>(v));  // row_bcast 15 into rows 1 and 3
    v = fmax(v, dpp_f64<0x143, 0xc>(v));  // row_bcast 31 into rows 2 and 3: lane 63 holds the wave's
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// workgroup barrier that orders LDS traffic only (a __syncthreads() also waits for the row's global stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(PS_T) void pchol_steps_kernel(const double* __restrict__ S, double* __restrict__ Y, int64_t mp, int jb,
                                                            int sb, int j0, int nsteps, double* __restrict__ dg,
                                                            PcholState* __restrict__ stt, int* __restrict__ order,
                                                            double* __restrict__ piv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    if (stt->done) return;
    double* Yl = reinterpret_cast<double*>(ps_smem);  // [PS_SUB][PS_T]: rows sb .. of the block
    double* hp = Yl + (size_t)PS_SUB * PS_T;          // [PS_SUB]: the pivot column's entries of the rows jb .. sb - 1
    double* sv = hp + PS_SUB;                          // [8] wave maxima
    int* si = reinterpret_cast<int*>(sv + 16);         // [8] their columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = tid;
    const bool live = i < mp;
    const int64_t ic = live ? i : 0;
    double di = live ? dg[i] : -INFINITY;
    const double tol = stt->tol;
    const int nold = sb - jb;  // rows of the block that are NOT LDS resident: 0 or PS_SUB
    double aold[PS_SUB];
#pragma unroll
    for (int u = 0; u < PS_SUB; ++u) aold[u] = (u < nold) ? Y[(int64_t)(jb + u) * mp + ic] : 0.0;
    for (int k = sb; k < j0; ++k) Yl[(k - sb) * PS_T + tid] = Y[(int64_t)k * mp + ic];  // (a launch that continues a sub-block)
    lds_barrier();
    int j = j0;
    for (int s = 0; s < nsteps; ++s) {
        // ---- p = argmax of the remaining diagonal, ties to the lowest column
        const double wm = wave_max_dpp(di);
        const unsigned long long hit = __ballot(di == wm);
        if (lane == 0) {
            sv[wave] = wm;
            si[wave] = wave * 64 + (int)__ffsll((long long)hit) - 1;
        }
        lds_barrier();
        double bv = sv[0];
        int bi = si[0];
#pragma unroll
        for (int w = 1; w < PS_T / 64; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) {
                bv = sv[w];
                bi = si[w];
            }
        if (!(bv > tol)) {  // uniform over the workgroup
            if (tid == 0) stt->done = 1;
            break;
        }
        const int64_t p = bi;
        // ---- the step's one global round trip: the pivot's row of S and (second half of a block) its column of the old rows
        if (wave == 0 && lane < nold) hp[lane] = Y[(int64_t)(jb + lane) * mp + p];
        double acc = S[p * mp + ic], accp = S[p * mp + p];
        const int cnt = j - sb;  // LDS-resident rows so far
        if (nold != 0) {
            lds_barrier();  // hp complete
#pragma unroll
            for (int u = 0; u < PS_SUB; ++u) {
                const double p0 = hp[u];
                acc = fma(-aold[u], p0, acc);
                accp = fma(-p0, p0, accp);
            }
        }
        {
            const double* yc = Yl + tid;
            const double* yp = Yl + p;
            int k = 0;
            for (; k + 4 <= cnt; k += 4) {
                const double p0 = yp[k * PS_T], p1 = yp[(k + 1) * PS_T], p2 = yp[(k + 2) * PS_T], p3 = yp[(k + 3) * PS_T];
                const double a0 = yc[k * PS_T], a1 = yc[(k + 1) * PS_T], a2 = yc[(k + 2) * PS_T], a3 = yc[(k + 3) * PS_T];
                acc = fma(-a0, p0, acc);
                accp = fma(-p0, p0, accp);
                acc = fma(-a1, p1, acc);
                accp = fma(-p1, p1, accp);
                acc = fma(-a2, p2, acc);
                accp = fma(-p2, p2, accp);
                acc = fma(-a3, p3, acc);
                accp = fma(-p3, p3, accp);
            }
            for (; k < cnt; ++k) {
                const double p0 = yp[k * PS_T], a0 = yc[k * PS_T];
                acc = fma(-a0, p0, acc);
                accp = fma(-p0, p0, accp);
            }
        }
        const bool used = !live || di == -INFINITY;
        const bool badp = !(accp > 0.25 * tol);  // a pivot the fresh evaluation does not confirm: retired with an all-zero row
        const double inv = badp ? 0.0 : 1.0 / sqrt(accp);
        const double y = (used || badp) ? 0.0 : acc * inv;
        Yl[cnt * PS_T + tid] = y;
        if (live) Y[(int64_t)j * mp + i] = y;
        di = (used || i == p) ? -INFINITY : di - y * y;
        if (tid == 0) {
            order[j] = (int)p;
            piv[j] = accp;
        }
        ++j;
        lds_barrier();  // row j is in LDS for the next step; sv / si / hp are free
    }
    if (live) dg[i] = di;
    if (tid == 0) stt->r = j;
}

// trailing update after a block of 64 pivot steps:  S -= Yb^T Yb,  Yb = rows jb .. jb + 63 of Y (64 x 64 tile per
// workgroup, f64 MFMA, both operand tiles staged k-major in LDS)
__global__ __launch_bounds__(256) void pchol_update_kernel(double* __restrict__ S, const double* __restrict__ Y,
                                                           int64_t mp, int jb, const PcholState* __restrict__ stt,
                                                           int need_panels) {
    // greedy blocks (need_panels < 0): skipped once the factorisation is finished; hint panels: run iff accepted
    if (need_panels < 0 ? stt->done != 0 : stt->panels != need_panels) return;
    __shared__ double sa[JP * LDK];
    __shared__ double sb[JP * LDK];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    double* pc = S + ((int64_t)ti * 64) * mp + (int64_t)tj * 64;
    f64x4 cin[2][2];  // this lane's entries of the S tile: in flight together with the panels (round 5 read them after the MFMA
                      // loop - a second exposed memory round trip per workgroup)
    {
        double va[16], vb[16];
        const double* ya = Y + (int64_t)jb * mp + (int64_t)ti * 64 + lane;
        const double* yb = Y + (int64_t)jb * mp + (int64_t)tj * 64 + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            va[q] = ya[(int64_t)(wave + 4 * q) * mp];
            vb[q] = yb[(int64_t)(wave + 4 * q) * mp];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cin[a][b][r] = pc[(int64_t)(wr + a * 16 + lk + 4 * r) * mp + wc + b * 16 + li];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            sa[(wave + 4 * q) * LDK + lane] = va[q];
            sb[(wave + 4 * q) * LDK + lane] = vb[q];
        }
    }
    __syncthreads();
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 4) {
        double fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fa[a] = sa[(kk + lk) * LDK + wr + a * 16 + li];  // A[i][k] = Yb[k][ti 64 + i]
            fb[a] = sb[(kk + lk) * LDK + wc + a * 16 + li];  // B[k][j] = Yb[k][tj 64 + j]
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr + a * 16 + lk + 4 * r;
                const int col = wc + b * 16 + li;
                pc[(int64_t)row * mp + col] = cin[a][b][r] - acc[a][b][r];
            }
}

// ---- pivot order of a NEARBY matrix as a hint: 64 pivots per 3 launches instead of 65 --------------------------------
// Between EM iterations the greedy pivot order barely moves (CPU prototype: 955 of 960 pivots of one iteration are
// acceptable, in order, for the next).  Any pivot order gives a valid factor A = L L^T + E with the same stopping rule;
// what pivoting must guarantee is only that no pivot is tiny against the diagonal entries still outside (rounding noise
// in a column is amplified by 1 / sqrt(pivot)).  So a panel of the previous order's next 64 columns is factored as a
// block - 64 x 64 Cholesky of the gathered Schur block (one workgroup), forward substitution of the 64 gathered rows of S
// (one lane per column of A), MFMA trailing update - and accepted column by column while pivot > max(theta dmax, tol),
// dmax = the largest remaining diagonal entry when the panel starts (theta = 2^-10: noise amplification <= 32).  The
// first rejected column ends the use of the hint; the greedy steps above finish the factorisation (and find any column
// the hint does not know).

// Right-looking Cholesky of the gathered 64 x 64 block, register tiled like potrf64 of mvf_solve.hip (thread (ty, tx) of a
// 16 x 16 grid keeps a[ty + 16 p][tx + 16 q], p >= q; the owners of a column publish it UNSCALED in its own LDS row, one
// barrier per column, Newton reciprocal of the pivot, unpredicated rank-1 update with the next column first), stopping at
// the first pivot that fails the threshold: columns 0 .. nvalid - 1 of Lu are then the accepted ones.
template <int JQ>
__device__ __forceinline__ bool panel_potrf_group(double (&r)[4][4], double (*Lu)[LDU], double (*Pub)[2][NB], double* dgs,
                                                  int& pb, int tx, int ty, double thr, int n, int& nvalid) {
    if (16 * JQ >= n) return false;
    if (tx < 2) {
#pragma unroll
        for (int p = JQ; p < 4; ++p) Pub[pb][tx][ty + 16 * p] = r[p][JQ];
        if (tx == 0) {
#pragma unroll
            for (int p = JQ; p < 4; ++p) Lu[16 * JQ][ty + 16 * p] = r[p][JQ];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int jx = 0; jx < 16; jx += 2) {  // two columns per step, as potrf64
        const int j = 16 * JQ + jx;
        if (j >= n) return false;
        const double* u0 = Pub[pb][0];
        const double* u1 = Pub[pb][1];
        const double d0 = u0[j], e = u0[j + 1], g = u1[j + 1];
        double c0r[4], c0c[4], c1r[4], c1c[4];
#pragma unroll
        for (int p = JQ; p < 4; ++p) {
            c0r[p] = u0[ty + 16 * p];
            c0c[p] = u0[tx + 16 * p];
            c1r[p] = u1[ty + 16 * p];
            c1c[p] = u1[tx + 16 * p];
        }
        if (!(d0 > thr && d0 <= 1.79e308)) {  // uniform: every thread reads the same pivot
            nvalid = j;
            return false;
        }
        const double t1 = fma(g, d0, -(e * e));  // 1 / d1 = d0 / (g d0 - e^2): two independent reciprocal chains (potrf64)
        const double inv0 = rcp_nr2(d0);
        const double inv1 = d0 * rcp_nr2(t1);
        const double d1 = t1 * inv0;
        const bool ok1 = j + 1 < n && d1 > thr && d1 <= 1.79e308;
        if (threadIdx.x == 0) {
            dgs[j] = d0;
            if (ok1) dgs[j + 1] = d1;
        }
        if (!ok1) {  // column j is accepted (it sits in Lu[j] since its publication), column j + 1 is not
            nvalid = min(j + 1, n);
            return false;
        }
        double l0[4], l1[4];
#pragma unroll
        for (int p = JQ; p < 4; ++p) {
            l0[p] = c0r[p] * inv0;
            c1r[p] = fma(-l0[p], e, c1r[p]);
            c1c[p] = fma(-(c0c[p] * inv0), e, c1c[p]);
            l1[p] = c1r[p] * inv1;
        }
#pragma unroll
        for (int p = JQ; p < 4; ++p) r[p][JQ] = fma(-l1[p], c1c[JQ], fma(-l0[p], c0c[JQ], r[p][JQ]));
        if (tx == jx + 1) {
#pragma unroll
            for (int p = JQ; p < 4; ++p) Lu[j + 1][ty + 16 * p] = c1r[p];
        }
        if (tx == jx + 2 || tx == jx + 3) {
            const int w = tx - jx - 2;
#pragma unroll
            for (int p = JQ; p < 4; ++p) Pub[pb ^ 1][w][ty + 16 * p] = r[p][JQ];
            if (w == 0) {
#pragma unroll
                for (int p = JQ; p < 4; ++p) Lu[j + 2][ty + 16 * p] = r[p][JQ];
            }
        }
#pragma unroll
        for (int q = JQ + 1; q < 4; ++q)
#pragma unroll
            for (int p = q; p < 4; ++p) r[p][q] = fma(-l1[p], c1c[q], fma(-l0[p], c0c[q], r[p][q]));
        __syncthreads();
        pb ^= 1;
    }
    return true;
}

__global__ __launch_bounds__(256) void pchol_panel_factor_kernel(const double* __restrict__ S, int64_t m, int64_t mp,
                                                                 const int* __restrict__ hint, int nh, int b,
                                                                 const double* __restrict__ dg, PcholState* __restrict__ stt,
                                                                 int* __restrict__ order, double* __restrict__ piv,
                                                                 double* __restrict__ Lcc, int* __restrict__ cand_out) {
    __shared__ double Lu[64][LDU];
    __shared__ double Pub[2][2][NB];
    __shared__ double dgs[64];
    __shared__ int cand[64];
    __shared__ double red[4];
    __shared__ double s_dmax;
    __shared__ int s_n;
    if (stt->done || stt->hint_broken || stt->panels != b) return;  // uniform
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int idx = 64 * b + tid;
        const int c = idx < nh ? hint[idx] : -1;
        cand[tid] = (c >= 0 && c < m && dg[c] != -INFINITY) ? c : -1;
    }
    double mx = -INFINITY;
    for (int64_t i = tid; i < mp; i += 256) mx = fmax(mx, dg[i]);
    const double t = -block_min<256>(-mx, red);  // has barriers: cand[] is visible afterwards
    if (tid == 0) {
        int n = 0;
        for (int j = 0; j < 64; ++j)
            if (cand[j] >= 0) cand[n++] = cand[j];
        for (int j = n; j < 64; ++j) cand[j] = -1;
        s_n = n;
        s_dmax = t;
    }
    __syncthreads();
    const int n = s_n;
    const double dmax = s_dmax, tol = stt->tol;
    if (n == 0 || !(dmax > tol)) {
        if (tid == 0) {
            if (!(dmax > tol))
                stt->done = 1;
            else
                stt->hint_broken = 1;
        }
        return;
    }
    const int tx = tid & 15, ty = tid >> 4;
    double r[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 16 * p, c = tx + 16 * q;
            r[p][q] = (i < n && c < n) ? S[(int64_t)cand[i] * mp + cand[c]] : (i == c ? 1.0 : 0.0);
        }
    const double thr = fmax(PCHOL_THETA * dmax, tol);
    int nvalid = n;
    int pb = 0;
    if (panel_potrf_group<0>(r, Lu, Pub, dgs, pb, tx, ty, thr, n, nvalid))
        if (panel_potrf_group<1>(r, Lu, Pub, dgs, pb, tx, ty, thr, n, nvalid))
            if (panel_potrf_group<2>(r, Lu, Pub, dgs, pb, tx, ty, thr, n, nvalid))
                panel_potrf_group<3>(r, Lu, Pub, dgs, pb, tx, ty, thr, n, nvalid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = tx + 16 * q;
        const double l = c < nvalid ? sqrt(dgs[c]) : 1.0;
        const double rl = 1.0 / l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = ty + 16 * p;
            double v = 0.0;
            if (i < nvalid && c <= i) v = c < i ? div_by(Lu[c][i], l, rl) : l;
            Lcc[i * 64 + c] = v;
        }
    }
    if (tid < 64) {
        cand_out[tid] = tid < nvalid ? cand[tid] : -1;
        order[64 * b + tid] = tid < nvalid ? cand[tid] : -1;
        if (tid < nvalid) piv[64 * b + tid] = dgs[tid];
    }
    if (tid == 0) {
        stt->panel_nvalid = nvalid;
        if (nvalid < n) stt->hint_broken = 1;
        if (nvalid > 0) stt->panels = b + 1;
    }
}

// rows 64 b .. 64 b + 63 of Y from the accepted panel: Y[64 b + j][i] = (S[c_j][i] - sum_{k < j} L[j][k] Y[64 b + k][i]) / L[j][j]
// (the panel's own pivot columns come out as L's columns and are forced to exact zeros behind their pivot), dg -= sum_j y_j^2,
// the workgroup's (max, argmax) for the greedy steps that may follow.  FOUR lanes (one DPP quad) per column i of A, each
// holding the 16 interleaved entries j = 4 jl + rho of that column's 64 - the substitution of trsm_panel_kernel
// (mvf_chol_dev.h: the same multiply-subtracts in the same order as the dot-product form, one lane per column, that this
// kernel used until round 5 - 29 us per launch behind a 2016-long dependent chain per lane).
__global__ __launch_bounds__(4 * PC_T) void pchol_panel_rows_kernel(const double* __restrict__ S, double* __restrict__ Y, int64_t mp,
                                                                    int b, double* __restrict__ dg, double* __restrict__ pm,
                                                                    PcholState* __restrict__ stt, const double* __restrict__ Lcc,
                                                                    const int* __restrict__ cand_in) {
    if (stt->panels != b + 1) return;
    __shared__ __align__(16) double Ls[NB * TLC];
    __shared__ double rd[64];
    __shared__ int cand[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nvalid = stt->panel_nvalid;
    {
        double v[8];  // L[i][c = lane], rows i = wave * 8 + q (Lcc: explicit zeros above the diagonal and behind nvalid)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = Lcc[(wave * 8 + q) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = wave * 8 + q;
            Ls[lane * TLC + (i & 3) * TLQ + (i >> 2)] = (lane < i) ? v[q] : 0.0;
        }
        if (tid < 64) {
            cand[tid] = cand_in[tid];
            rd[tid] = tid < nvalid ? 1.0 / Lcc[tid * 64 + tid] : 0.0;
        }
    }
    __syncthreads();
    const int rho = tid & 3;
    const int64_t i = (int64_t)blockIdx.x * PC_T + (tid >> 2);
    const bool live = i < mp;
    const int64_t ic = live ? i : 0;
    const double di = dg[ic];
    const bool used = !live || di == -INFINITY;
    int pos = -1;
    for (int k = 0; k < nvalid; ++k)
        if (cand[k] == (int)ic) pos = k;
    double y[16];
#pragma unroll
    for (int jl = 0; jl < 16; ++jl) {
        const int j = 4 * jl + rho;
        y[jl] = (j < nvalid && !used) ? S[(int64_t)max(cand[j], 0) * mp + ic] : 0.0;
    }
    const double* ls_rho = Ls + rho * TLQ;
    double2 first[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) first[h] = reinterpret_cast<const double2*>(ls_rho)[h];
    trsm_steps<0>(y, ls_rho, rd, first, rd[0]);
    double ss = 0.0;
#pragma unroll
    for (int jl = 0; jl < 16; ++jl) {
        const int j = 4 * jl + rho;
        double v = y[jl] * rd[j];
        if (pos >= 0 && j > pos) v = 0.0;
        y[jl] = v;
        ss = fma(v, v, ss);
    }
    // the quad's four partial sums in lane order (deterministic), on every lane of the quad
    {
        const double s0 = quad_bcast<0>(ss), s1 = quad_bcast<1>(ss), s2 = quad_bcast<2>(ss), s3 = quad_bcast<3>(ss);
        ss = ((s0 + s1) + s2) + s3;
    }
    const double dn = (used || pos >= 0) ? -INFINITY : di - ss;
    if (live) {
#pragma unroll
        for (int jl = 0; jl < 16; ++jl) Y[((int64_t)64 * b + 4 * jl + rho) * mp + i] = y[jl];
        if (rho == 0) dg[i] = dn;
    }
    __shared__ double sv[4 * PC_T / 64];
    __shared__ int si[4 * PC_T / 64];
    double wv = live ? dn : -INFINITY;
    int wi = live ? (int)i : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(wv, o, 64);
        const int ix = __shfl_xor(wi, o, 64);
        if (v > wv || (v == wv && ix < wi)) {
            wv = v;
            wi = ix;
        }
    }
    if (lane == 0) {
        sv[wave] = wv;
        si[wave] = wi;
    }
    __syncthreads();
    if (tid == 0) {
        double v = sv[0];
        int ix = si[0];
#pragma unroll
        for (int w = 1; w < 4 * PC_T / 64; ++w)
            if (sv[w] > v || (sv[w] == v && si[w] < ix)) {
                v = sv[w];
                ix = si[w];
            }
        pm[2 * blockIdx.x] = v;
        pm[2 * blockIdx.x + 1] = (double)ix;
        if (blockIdx.x == 0) stt->r = 64 * b + nvalid;  // only the last accepted panel can be partial
    }
}

// ================= deflated truncated solve: the gelsd cut-off without the full eigendecomposition ======================
// The pivoted factor A = L L^T (r columns, rows of Y) has dropped everything below tolf eps lambda_max already, so the
// eigenvalues of L L^T that gelsd truncates (<= eps lambda_max) are the FEW smallest ones of the r x r matrix S2 = L^T L -
// about a tenth of r in the EM's systems - and they sit within 1 / tolf of the cut.  Instead of orthogonalising all r
// columns (Jacobi, ~13 sweeps) only that invariant subspace is computed:
//   S2 = Rc Rc^T (Cholesky: accurate although cond(S2) ~ 1 / (tolf eps), because the pivoted factor is graded), the
//   inverse factor rides along as extra rows of the same factorisation, Minv = S2^-1;
//   block inverse iteration on DEFL_B vectors started on the trailing (smallest pivot) rows: Z <- orth(Minv Z), two
//   applications; Rayleigh-Ritz H = Z S2 Z^T (DEFL_B x DEFL_B, condition ~ 1e2: a small well-conditioned eigenproblem for
//   the Jacobi kernels above); W = the Ritz vectors with theta <= rcond lambda_max, Pc = I - W^T W.
//   C = L Pc Minv Pc Minv Pc L^T R   - the projections BETWEEN the two inverse applications are what keeps the
//   amplified rounding error of the dropped directions out of the result (subtracting their terms afterwards cancels
//   catastrophically).  tools/lrproto_partial2.py: field within 4e-7 ... 2e-6 of the exact truncated solve.
// Falls back to the Jacobi path when more than DEFL_B - DEFL_GUARD Ritz values lie below the cut, when r < 2 DEFL_B or a
// factorisation meets a non-positive pivot.
constexpr int DEFL_B = 256;
constexpr int DEFL_GUARD = 32;
constexpr int DEFL_SMALL_MAX = 72;     // the previous call truncated at most this many: try a 128-vector block first
constexpr int DEFL_SMALL_ACCEPT = 80;  // ... and accept its result only if it finds at most this many
// Small factors (2 DEFL_TINY <= r < 2 DEFL_B: M = 128 .. 640 control points, BASELINE configs 2 and 5): a 64-vector block and
// three applications.  Measured spectra at M = 500 (50 k and 250 k cells, lambda_ = 0.02): 1 - 2 eigenvalues below the cut,
// the 65th at 2^10 x the cut - the block converges in one application; accepted while at most DEFL_TINY_ACCEPT lie below.
constexpr int DEFL_TINY = 64;
constexpr int DEFL_TINY_ACCEPT = 40;

struct DeflBuf {
    size_t s2, cw, za, zb, wsel, g, h, yh, cwb, ta, tb, cb, dummy, part, theta, total;
};

static DeflBuf defl_layout(int64_t rp) {
    DeflBuf d;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    const size_t b = DEFL_B;
    d.s2 = take((size_t)rp * rp * sizeof(double));
    d.cw = take(chol_inv_workspace_bytes(rp));
    d.za = take(b * rp * sizeof(double));
    d.zb = take(b * rp * sizeof(double));
    d.wsel = take(b * rp * sizeof(double));
    d.g = take(b * b * sizeof(double));
    d.h = take(b * b * sizeof(double));
    d.yh = take(b * b * sizeof(double));
    d.cwb = take(chol_inv_workspace_bytes(b));
    d.ta = take((size_t)rp * 8 * sizeof(double));
    d.tb = take((size_t)rp * 8 * sizeof(double));
    d.cb = take(b * 8 * sizeof(double));
    d.dummy = take((size_t)std::max<int64_t>(rp, b) * 8 * sizeof(double));
    d.part = take((size_t)16 * rp * 8 * sizeof(double));
    d.theta = take(b * sizeof(double));
    d.total = o;
    return d;
}

static size_t defl_scratch_bytes(int64_t mp) { return defl_layout(mp).total; }

// theta_i = ||Yh_i||^2 (rows of the orthogonalised factor of H = sigma_i u_i^T).  Rows with theta <= rcond lambda_max become
// u_i^T (the directions to deflate), the others zero.  Diagnostics in einfo's layout (see jac_scale_kernel); einfo[4] = the
// number of deflated directions.
__global__ __launch_bounds__(256) void defl_select_kernel(const double* __restrict__ theta, int b,
                                                          const PcholState* __restrict__ stt, double rcond, int64_t r,
                                                          double* __restrict__ Yh, double* __restrict__ einfo) {
    __shared__ double red[4];
    const double lmax = stt->lmax_est, cut = rcond * lmax;
    {   // this workgroup's rows: blockIdx.x * 4 + wave
        const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i < b) {
            const double t = theta[i];
            const double g = (!(t > cut) && t > 0.0) ? 1.0 / sqrt(t) : 0.0;
            for (int c = threadIdx.x & 63; c < b; c += 64) Yh[(int64_t)i * b + c] *= g;
        }
    }
    if (blockIdx.x != 0) return;
    double nsel = 0.0, mink = INFINITY, minl = INFINITY;
    for (int i = threadIdx.x; i < b; i += 256) {
        const double t = theta[i];
        const bool sel = !(t > cut);
        nsel += sel ? 1.0 : 0.0;
        if (!sel) mink = fmin(mink, t);
        minl = fmin(minl, t);
    }
    const double k1 = block_sum<256>(nsel, red);
    const double k2 = block_min<256>(mink, red);
    const double k3 = block_min<256>(minl, red);
    if (threadIdx.x == 0) {
        einfo[1] = (double)r - k1;
        einfo[2] = lmax;
        einfo[3] = k2;
        einfo[4] = k1;
        einfo[5] = k3;
    }
}

// T[n][d] -= sum over the splits of part[split][n][d]   (the second half of  T -= W^T (W T))
__global__ __launch_bounds__(256) void defl_sub_kernel(const double* __restrict__ part, int nsplit, int64_t n,
                                                       double* __restrict__ T) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * 8) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += part[(int64_t)q * n * 8 + e];
    T[e] -= s;
}

// ---- the direct form of the small deflated solve (round 5) ---------------------------------------------------------------
// When the previous call on the workspace factored ALL m columns (factor rank r = m: what M <= 640 control points give, BASELINE
// configs 2 and 5) the pivoted Cholesky of the next, nearby matrix is, in that pivot order, an ordinary Cholesky of the permuted
// matrix: A_perm = Rc Rc^T.  The blocked factorisation with the inverse riding along then yields A_perm^-1 = Rc^-T Rc^-1 in
// ONE factorisation - no pivoted factor, no S2 = L^T L and no second factorisation - and the truncated solve is
//     C = Pi^T Pc E E^T Pc Pi R,   E = Rc^-T,  Pc = I - W^T W,  W = the eigenvectors of A_perm with lambda <= rcond lambda_max
// (block inverse iteration on 64 vectors + Rayleigh-Ritz as in the factor form; the projection BEFORE the inverse keeps the
// dropped directions' 1 / lambda out, the one after it removes what rounding leaked back).  Two things keep the GRADING of the
// factor, without which the field loses 3 - 4 digits (measured: the product E E^T applied once, and H = Z A_perm Z^T, agree
// with the factor form to 1e-6 ... 8e-3 only): the inverse is applied as its two triangular factors, never as their product,
// and the Rayleigh-Ritz matrix is the Gram matrix of B = Z Rc (H = B B^T), not a product with the assembled matrix.  Accepted only if every
// pivot clears the stopping tolerance tolf eps lambda_max (the factor form would have kept all m columns too) and at most
// DEFL_TINY_ACCEPT Ritz values lie below the cut; anything else re-runs the call in the factor form.
__global__ __launch_bounds__(256) void direct_prepare_kernel(const double* __restrict__ S, int64_t m, int64_t mp, double tolf,
                                                             PcholState* __restrict__ stt, int* __restrict__ info,
                                                             int* __restrict__ dflag, int form_hint) {
    __shared__ double red[4];
    double mx = 0.0;
    bool finite = true;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        const double d = S[i * mp + i];
        finite = finite && (fabs(d) <= 1.79e308);
        mx = fmax(mx, d);
    }
    const double bad = block_sum<256>(finite ? 0.0 : 1.0, red);
    const double t = -block_min<256>(-mx, red);
    if (threadIdx.x == 0) {
        const double est = stt->lmax_est;
        const bool ok = bad == 0.0 && (fabs(est) <= 1.79e308);
        const double lmax = fmax(ok ? est : 0.0, t);
        stt->maxdiag = t;
        stt->lmax_est = lmax;
        stt->tol = tolf * 2.220446049250313e-16 * lmax;
        info[0] = ok ? 0 : 1;
        // the pivot order in the workspace must be that of a finished factorisation of ALL m columns (magic / order_len stay
        // untouched: the factor form, if it has to answer, reads them for its own hint)
        int valid = (ok && stt->magic == PCHOL_MAGIC_V && stt->order_len == (int)m) ? 1 : 0;
        // the asynchronous entry point (form_hint != 0) took the host's word for what the previous call left here and has
        // launched accordingly: the state must really be that (no cool-down pending; 2 = a complete direct-form call)
        if (form_hint != 0 && stt->direct_skip != 0) valid = 0;
        if (form_hint == 2 && !(stt->defl == 2 && stt->pad4 == 1 && stt->defl_block == DEFL_TINY)) valid = 0;
        dflag[0] = valid;
        // the kept Rayleigh-Ritz rotation is that of the previous call only if that call was answered by the direct form
        dflag[1] = (valid && stt->defl == 2 && stt->pad4 == 1) ? 1 : 0;
    }
}

// piv[j] = Rc_jj^2 (what the pivoted factorisation records); the direct form is void unless every pivot clears the tolerance
__global__ __launch_bounds__(256) void direct_check_kernel(const double* __restrict__ rdiag, int64_t m,
                                                           const PcholState* __restrict__ stt, const int* __restrict__ info,
                                                           double* __restrict__ piv, int* __restrict__ dflag) {
    __shared__ double red[4];
    double lo = INFINITY;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        const double l = 1.0 / rdiag[i];
        const double v = l * l;
        piv[i] = v;
        lo = (v == v) ? fmin(lo, v) : -INFINITY;  // NaN pivots fail the test
    }
    const double t = block_min<256>(lo, red);
    if (threadIdx.x == 0 && (!(t >= stt->tol) || info[0] != 0)) dflag[0] = 0;
}

// Lo = the lower triangle of the factorisation's work matrix (its blocks above the diagonal still hold matrix entries);
// Et = E^T (E = Rc^-T, upper triangular): both n x n with leading dimension n
__global__ __launch_bounds__(256) void tril_and_transpose_kernel(const double* __restrict__ W, const double* __restrict__ E,
                                                                 int64_t n, double* __restrict__ Lo, double* __restrict__ Et) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    Lo[i * n + j] = j <= i ? W[i * n + j] : 0.0;
    Et[i * n + j] = E[j * n + i];
}

// rep[0..5] = einfo[0..5], rep[6] = info, rep[7] = the state flag, rep[8..10] = lambda_max estimate, the one before, max diagonal
__global__ void direct_report_kernel(const double* __restrict__ einfo, const int* __restrict__ info,
                                     const int* __restrict__ dflag, const PcholState* __restrict__ stt,
                                     double* __restrict__ rep) {
    const int t = threadIdx.x;
    if (t < 6) rep[t] = einfo[t];
    if (t == 6) rep[6] = (double)info[0];
    if (t == 7) rep[7] = (double)dflag[0];
    if (t == 8) rep[8] = stt->lmax_est;
    if (t == 9) rep[9] = stt->lmax_prev;
    if (t == 10) rep[10] = stt->maxdiag;
}

// the closing state of a direct-form call: a finished factorisation of all m columns in the unchanged pivot order, defl = 2
// (this form), the block size, pad4 = 1 (block and Rayleigh-Ritz rotation are this call's complete ones) and the count of
// deflated directions; einfo[0] = Rayleigh-Ritz launches, [4] = 0 (no shift), [6] = r = m, [7] = block
__global__ void direct_finish_kernel(PcholState* __restrict__ stt, double* __restrict__ einfo, int m, int b, int hsweeps) {
    const int nsel = (int)einfo[4];
    stt->done = 1;
    stt->r = m;
    stt->magic = PCHOL_MAGIC_V;
    stt->order_len = m;
    stt->keep_len = m;
    stt->defl = 2;
    stt->defl_block = b;
    stt->pad4 = hsweeps == 1 ? 1 : 0;
    stt->defl_nsel = nsel;
    stt->direct_skip = 0;
    einfo[0] = (double)hsweeps;
    einfo[4] = 0.0;
    einfo[6] = (double)m;
    einfo[7] = (double)b;
    einfo[8] = 2.0;  // answered by the direct form
    einfo[9] = 0.0;
}

// mvf_solve_minnorm_lrd_async: the acceptance test of the direct form ON THE DEVICE (the synchronous entry point copies
// the same eleven numbers to the host and decides there) + the closing state.  Not accepted: einfo[9] = 1, the workspace
// keeps its pivot order and gets the cool-down mark, so that the caller's repeat through mvf_solve_minnorm_lrd goes to the
// factor form at once.
__global__ void direct_close_kernel(PcholState* __restrict__ stt, double* __restrict__ einfo, const int* __restrict__ info,
                                    const int* __restrict__ dflag, const unsigned int* __restrict__ rot, int m, int b,
                                    double accept_n) {
    const double q_est = stt->lmax_est, q_prev = stt->lmax_prev;
    const bool lmax_ok = fabs(q_est - q_prev) <= 1e-7 * q_est || q_est == stt->maxdiag;
    const bool ok = info[0] == 0 && dflag[0] == 1 && rot[0] == 0u && lmax_ok && einfo[4] <= accept_n &&
                    fabs(einfo[5]) <= 1.79e308 && einfo[5] > 0.0;
    if (ok) {
        const int nsel = (int)einfo[4];
        stt->done = 1;
        stt->r = m;
        stt->magic = PCHOL_MAGIC_V;
        stt->order_len = m;
        stt->keep_len = m;
        stt->defl = 2;
        stt->defl_block = b;
        stt->pad4 = 1;
        stt->defl_nsel = nsel;
        stt->direct_skip = 0;
        einfo[0] = 1.0;
        einfo[4] = 0.0;
        einfo[6] = (double)m;
        einfo[7] = (double)b;
        einfo[8] = 2.0;
        einfo[9] = 0.0;
    } else {
        if (stt->direct_skip == 0) stt->direct_skip = 4;  // (a cool-down that is already running keeps its count)
        einfo[8] = 2.0;
        einfo[9] = 1.0;
    }
}

// T[i][0..8) = R[order[i]][0..nrhs) (zero padded), and back: C[order[i]][d] = T[i][d]
__global__ __launch_bounds__(256) void perm_rows_kernel(const double* __restrict__ R, int nrhs, const int* __restrict__ order,
                                                        int64_t m, int64_t rp, const int* __restrict__ dflag,
                                                        double* __restrict__ T) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rp * 8) return;
    const int64_t i = e >> 3;
    const int d = (int)(e & 7);
    double v = 0.0;
    if (i < m && d < nrhs) {
        int64_t oi = dflag[0] ? order[i] : i;
        if (oi < 0 || oi >= m) oi = i;
        v = R[oi * nrhs + d];
    }
    T[e] = v;
}
__global__ __launch_bounds__(256) void unperm_rows_kernel(const double* __restrict__ T, int nrhs, const int* __restrict__ order,
                                                          int64_t m, const int* __restrict__ dflag, double* __restrict__ C) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * 8) return;
    const int64_t i = e >> 3;
    const int d = (int)(e & 7);
    if (d >= nrhs) return;
    int64_t oi = dflag[0] ? order[i] : i;
    if (oi < 0 || oi >= m) oi = i;
    C[oi * nrhs + d] = T[e];
}

struct LrPlan {
    int64_t mp;
    int nbmax, npmax, nwg, spart_tiles;
    size_t off_s, off_y, off_dg, off_pm, off_x, off_state, off_order, off_piv, off_spart, off_j, off_flags, off_stamps,
        off_sig2, off_t, off_part, off_rot, off_scal, off_hint, off_lcc, off_cand, off_xkeep, off_dflag, off_jkeep, total, off_d, total_d;
};

constexpr int LR_GRAM_WGS = 512;  // upper bound of the workgroups per Jacobi Gram launch (pairs x K splits)
constexpr int LR_GRAM_WGS_DEFAULT = 256;  // measured at m = 3000, r = 860: 512 -> 21.3, 256 -> 19.9, 128 -> 21.5 ms of Jacobi

static LrPlan lr_plan(int64_t m) {
    LrPlan p;
    p.mp = cdiv(m, 64) * 64;
    p.nbmax = (int)(p.mp / JB);
    p.npmax = p.nbmax / 2;
    p.nwg = (int)cdiv(p.mp, PC_T);
    p.spart_tiles = LR_GRAM_WGS + p.npmax;  // >= npairs * nsplit for every rank
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    p.off_s = take((size_t)p.mp * p.mp * sizeof(double));
    p.off_y = take((size_t)p.mp * p.mp * sizeof(double));
    p.off_dg = take((size_t)2 * p.mp * sizeof(double));
    p.off_pm = take((size_t)4 * p.nwg * sizeof(double));
    p.off_x = take((size_t)2 * p.mp * sizeof(double));
    p.off_state = take(sizeof(PcholState));
    p.off_order = take((size_t)p.mp * sizeof(int));
    p.off_piv = take((size_t)p.mp * sizeof(double));
    p.off_spart = take((size_t)p.spart_tiles * JP * JP * sizeof(double));
    p.off_j = take((size_t)p.npmax * JP * JP * sizeof(double));
    p.off_flags = take((size_t)p.npmax * sizeof(int));
    p.off_stamps = take((size_t)(p.nbmax + (size_t)p.nbmax * p.nbmax) * sizeof(int));
    p.off_sig2 = take((size_t)p.mp * sizeof(double));
    p.off_t = take((size_t)p.mp * 8 * sizeof(double));
    p.off_part = take((size_t)16 * m * 8 * sizeof(double));
    p.off_rot = take(256);
    p.off_scal = take(256);
    p.off_hint = take((size_t)p.mp * sizeof(int));
    p.off_lcc = take((size_t)64 * 64 * sizeof(double));
    p.off_cand = take(256);
    p.off_xkeep = take((size_t)p.mp * sizeof(double));
    p.off_dflag = take(256);
    p.off_jkeep = take((size_t)JP * JP * sizeof(double));
    p.total = o;
    // scratch of the deflated solve (mvf_solve_minnorm_lrd only; sized for a factor of full width)
    p.off_d = take(defl_scratch_bytes(p.mp));
    p.total_d = o;
    return p;
}


// ---- diag(U pinv(A) U^T) from the decomposition a solve left in its workspace ------------------------------------------
// (the alignment's variational sigma^2 needs SigmaDiag = sigma2 diag(U pinv(SigmaInv) U^T), morpho_class.py:1295-1297).
// Rows of Y are sigma_i w_i^T, so  pinv(A) = sum_kept w_i w_i^T / lambda_i  and
//   d_n = sum_i g_i (sum_m K(x_n, c_m) Y[i][m])^2,   g_i = [kept] / (sigma_i^2 lambda_i)   (the weights of the solve).
__global__ __launch_bounds__(256) void pinv_weights_kernel(const double* __restrict__ sig2, int64_t nrows,
                                                           const double* __restrict__ scal, double rcond,
                                                           double* __restrict__ gw) {
    __shared__ double red[4];
    __shared__ double bc;
    const double delta = scal[1];
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < nrows; i += 256)
        if (sig2[i] > 0.0) mx = fmax(mx, fabs(sig2[i] - delta));
    const double t = -block_min<256>(-mx, red);
    if (threadIdx.x == 0) bc = t;
    __syncthreads();
    const double cut = rcond * bc;
    for (int64_t i = threadIdx.x; i < nrows; i += 256) {
        const double s2 = sig2[i], lam = s2 - delta;
        gw[i] = (s2 > 0.0 && fabs(lam) > cut) ? 1.0 / (s2 * lam) : 0.0;
    }
}

constexpr int PD_CT = 128;  // control points per LDS stage
constexpr int PD_RB = 8;    // rows of Y per pass over the control points
template <typename T>
__global__ __launch_bounds__(256) void pinv_diag_kernel(const T* __restrict__ x4, int64_t n, const T* __restrict__ ctrl4,
                                                        int64_t m, T s, const double* __restrict__ Y, int64_t nrows,
                                                        int64_t mp, const double* __restrict__ gw,
                                                        double* __restrict__ out) {
    using V4T = typename Vec4<T>::type;
    __shared__ V4T sc[PD_CT];
    __shared__ double sy[PD_RB][PD_CT];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const V4T xv = (i < n) ? reinterpret_cast<const V4T*>(x4)[i] : V4T{0, 0, 0, 0};
    const T px = xv.x * s, py = xv.y * s, pz = xv.z * s;
    double d = 0.0;
    for (int64_t rb = 0; rb < nrows; rb += PD_RB) {
        double acc[PD_RB];
#pragma unroll
        for (int q = 0; q < PD_RB; ++q) acc[q] = 0.0;
        for (int64_t m0 = 0; m0 < m; m0 += PD_CT) {
            const int mc = (int)min((int64_t)PD_CT, m - m0);
            __syncthreads();
            if (threadIdx.x < PD_CT) {
                const int j = threadIdx.x;
                if (j < mc) {
                    const V4T cv = reinterpret_cast<const V4T*>(ctrl4)[m0 + j];
                    sc[j] = V4T{cv.x * s, cv.y * s, cv.z * s, 0};
                } else {
                    sc[j] = V4T{0, 0, 0, 0};
                }
            }
            for (int e = threadIdx.x; e < PD_RB * PD_CT; e += 256) {
                const int q = e / PD_CT, j = e % PD_CT;
                sy[q][j] = (rb + q < nrows && j < mc) ? Y[(rb + q) * mp + m0 + j] : 0.0;  // zero rows / columns add nothing
            }
            __syncthreads();
            for (int j = 0; j < mc; ++j) {
                const V4T cv = sc[j];
                const double k = (double)kernel_value(px, py, pz, cv.x, cv.y, cv.z);
#pragma unroll
                for (int q = 0; q < PD_RB; ++q) acc[q] = fma(k, sy[q][j], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < PD_RB; ++q)
            if (rb + q < nrows) d = fma(gw[rb + q] * acc[q], acc[q], d);
    }
    if (i < n) out[i] = d;
}

struct JacPlan {
    int64_t mp;
    int nb, npairs, nsplit, kchunks, bsplit, rows_per_split;
    size_t off_y, off_aux, off_spart, off_j, off_flags, off_stamps, off_sig2, off_t, off_part, off_rot, total;
};

static JacPlan jac_plan(int64_t m, int nrhs) {
    JacPlan p;
    p.mp = cdiv(m, 64) * 64;
    p.nb = (int)(p.mp / JB);
    p.npairs = p.nb / 2;
    const int nk = (int)(p.mp / 64);
    const int target = 512;  // workgroups per Gram launch
    int want = std::max(1, target / p.npairs);
    p.nsplit = std::min(nk, want);
    p.kchunks = (int)cdiv(nk, p.nsplit);
    p.nsplit = (int)cdiv(nk, p.kchunks);
    p.bsplit = (int)std::min<int64_t>(16, cdiv(p.mp, 64));
    p.rows_per_split = (int)cdiv(p.mp, p.bsplit);
    size_t o = align_up(chol_workspace_bytes(m, 1), 256);
    p.off_y = o;
    o += align_up((size_t)p.mp * p.mp * sizeof(double), 256);
    p.off_aux = o;  // warm start: assembled matrix / transformed matrix / back-transformed factor
    o += align_up((size_t)p.mp * p.mp * sizeof(double), 256);
    p.off_spart = o;
    o += align_up((size_t)p.npairs * p.nsplit * JP * JP * sizeof(double), 256);
    p.off_j = o;
    o += align_up((size_t)p.npairs * JP * JP * sizeof(double), 256);
    p.off_flags = o;
    o += align_up((size_t)p.npairs * sizeof(int), 256);
    p.off_stamps = o;  // mod[nb] | clean[nb * nb]
    o += align_up((size_t)(p.nb + (size_t)p.nb * p.nb) * sizeof(int), 256);
    p.off_sig2 = o;
    o += align_up((size_t)p.mp * sizeof(double), 256);
    p.off_t = o;
    o += align_up((size_t)p.mp * 8 * sizeof(double), 256);
    p.off_part = o;
    o += align_up((size_t)p.bsplit * m * 8 * sizeof(double), 256);
    p.off_rot = o;
    o += 256;
    p.total = o;
    return p;
}

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_solve_minnorm_workspace_bytes(int64_t m, int nrhs) {
    if (m <= 0) return 0;
    return jac_plan(m, nrhs).total;
}

extern "C" size_t mvf_solve_minnorm_basis_bytes(int64_t m) {
    if (m <= 0) return 0;
    const int64_t mp = cdiv(m, 64) * 64;
    return (size_t)mp * mp * sizeof(double);
}

extern "C" int mvf_solve_minnorm(const double* G, const double* K, double lambda_sigma2, double shift, double rcond,
                                 const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo,
                                 int max_sweeps, int reuse, double* basis, int warm, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    MVF_REQUIRE(m >= 0 && nrhs >= 1 && nrhs <= 8, "mvf_solve_minnorm: need m >= 0 and 1 <= nrhs <= 8 (got m=%lld nrhs=%d)",
                (long long)m, nrhs);
    MVF_REQUIRE(info && einfo, "mvf_solve_minnorm: null info / einfo");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
        return 0;
    }
    MVF_REQUIRE(G && K && R && C, "mvf_solve_minnorm: null pointer");
    MVF_REQUIRE(std::isfinite(lambda_sigma2) && lambda_sigma2 >= 0.0 && shift > 0.0 && shift < 1.0 && rcond >= 0.0,
                "mvf_solve_minnorm: bad regularisation / shift / rcond");
    if (max_sweeps <= 0) max_sweeps = 60;
    const JacPlan p = jac_plan(m, nrhs);
    MVF_REQUIRE(workspace && workspace_bytes >= p.total, "mvf_solve_minnorm: workspace too small (%zu < %zu)",
                workspace_bytes, p.total);
    char* ws = (char*)workspace;
    double* Y = (double*)(ws + p.off_y);
    double* Spart = (double*)(ws + p.off_spart);
    double* Jbuf = (double*)(ws + p.off_j);
    int* flags = (int*)(ws + p.off_flags);
    double* sig2 = (double*)(ws + p.off_sig2);
    double* T = (double*)(ws + p.off_t);
    double* part = (double*)(ws + p.off_part);
    unsigned int* rot = (unsigned int*)(ws + p.off_rot);

    const int64_t mp = p.mp;
    if (reuse) {
        // the workspace still holds the orthogonalised factor (Y, sig2) and delta of the previous call for this matrix
        CholPlan cq;
        chol_layout(m, 0, workspace, &cq);
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(mp, 4)), dim3(256), 0, st, Y, mp, mp, m, R, nrhs, sig2, T);
        hipLaunchKernelGGL(jac_scale_kernel, dim3(1), dim3(256), 0, st, sig2, mp, cq.scal, rcond, T, einfo + 6);
        hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)cdiv(mp, 64), (unsigned)p.bsplit), dim3(256), 0, st, Y, mp, mp, m,
                           T, p.rows_per_split, part);
        hipLaunchKernelGGL(jac_back_reduce_kernel, dim3((unsigned)cdiv(m * nrhs, 256)), dim3(256), 0, st, part,
                           p.bsplit, m, nrhs, C);
        MVF_LAUNCH_CHECK();
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
        return 0;
    }
    // 1. A + delta I = L L^T (no right-hand sides ride along); warm: of A' = Wt A Wt^T
    CholPlan cp;
    double* aux = (double*)(ws + p.off_aux);
    MVF_REQUIRE(!warm || basis, "mvf_solve_minnorm: warm start without a basis");
    if (warm) {
        hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mp), dim3(256), 0, st, G, K,
                           lambda_sigma2, m, mp, aux);
        gemm<false, true>(st, aux, mp, basis, mp, Y, mp, mp, mp, mp);   // T1 = A Wt^T
        gemm<false, false>(st, basis, mp, Y, mp, aux, mp, mp, mp, mp);  // A' = Wt T1
        MVF_LAUNCH_CHECK();
        if (int rc = chol_factor_mat(st, aux, mp, shift, m, workspace, &cp, info)) return rc;
    } else if (int rc = chol_factor(st, G, K, lambda_sigma2, shift, nullptr, m, 0, workspace, &cp, info)) {
        return rc;
    }
    int hinfo = 0;
    MVF_CHECK_HIP(rb_copy(st, &hinfo, info, sizeof(int)));
    MVF_CHECK_HIP(rb_sync(st));
    if (hinfo != 0) return 0;  // shift too small for this matrix: info[0] tells the caller, who escalates it
    MVF_REQUIRE(cp.mp == p.mp, "mvf_solve_minnorm: internal padding mismatch");

    // 2. one-sided block Jacobi on the columns of L
    hipLaunchKernelGGL(jac_init_kernel, dim3((unsigned)(mp / 64), (unsigned)(mp / 64)), dim3(256), 0, st, cp.W, m, mp, Y);
    MVF_LAUNCH_CHECK();
    const double tol = std::sqrt((double)m) * 2.220446049250313e-16;
    int* mod = (int*)(ws + p.off_stamps);
    int* clean = mod + p.nb;
    MVF_CHECK_HIP(hipMemsetAsync(mod, 0, (size_t)(p.nb + (size_t)p.nb * p.nb) * sizeof(int), st));  // all pairs dirty
    int sweeps = 0;
    unsigned int hrot = 1;
    while (sweeps < max_sweeps) {
        MVF_CHECK_HIP(hipMemsetAsync(rot, 0, sizeof(unsigned int), st));
        for (int r = 0; r < p.nb - 1; ++r) {
            const int stamp = 1 + sweeps * (p.nb - 1) + r;
            hipLaunchKernelGGL(jac_gram_kernel, dim3((unsigned)p.npairs, (unsigned)p.nsplit), dim3(256), 0, st, Y, mp,
                               p.nb, r, p.nsplit, p.kchunks, mod, clean, Spart);
            if (r == 0)
                hipLaunchKernelGGL(jac_eig_kernel<true>, dim3((unsigned)p.npairs), dim3(EIG_THREADS), 0, st, Spart, p.nsplit, tol,
                                   p.nb, r, stamp, mod, clean, Jbuf, flags, rot);
            else
                hipLaunchKernelGGL(jac_eig_kernel<false>, dim3((unsigned)p.npairs), dim3(EIG_THREADS), 0, st, Spart, p.nsplit,
                                   tol, p.nb, r, stamp, mod, clean, Jbuf, flags, rot);
            hipLaunchKernelGGL(jac_update_kernel, dim3((unsigned)p.npairs, (unsigned)(mp / 64)), dim3(256), 0, st, Y, mp,
                               p.nb, r, Jbuf, flags);
        }
        MVF_LAUNCH_CHECK();
        ++sweeps;
        MVF_CHECK_HIP(rb_copy(st, &hrot, rot, sizeof(unsigned int)));
        MVF_CHECK_HIP(rb_sync(st));
        if (hrot == 0) break;
    }

    if (warm) {
        // back to the original coordinates: row i of Y (= sigma_i x eigenvector i of A') -> Y Wt
        gemm<false, false>(st, Y, mp, basis, mp, aux, mp, mp, mp, mp);
        MVF_LAUNCH_CHECK();
        MVF_CHECK_HIP(hipMemcpyAsync(Y, aux, (size_t)mp * mp * sizeof(double), hipMemcpyDeviceToDevice, st));
    }

    // 3. truncated minimum-norm solve
    hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(mp, 4)), dim3(256), 0, st, Y, mp, mp, m, R, nrhs, sig2, T);
    hipLaunchKernelGGL(jac_scale_kernel, dim3(1), dim3(256), 0, st, sig2, mp, cp.scal, rcond, T, einfo);
    hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)cdiv(mp, 64), (unsigned)p.bsplit), dim3(256), 0, st, Y, mp, mp, m, T,
                       p.rows_per_split, part);
    hipLaunchKernelGGL(jac_back_reduce_kernel, dim3((unsigned)cdiv(m * nrhs, 256)), dim3(256), 0, st, part, p.bsplit, m,
                       nrhs, C);
    if (basis)
        hipLaunchKernelGGL(basis_extract_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mp), dim3(256), 0, st, Y, sig2,
                           mp, basis);
    MVF_LAUNCH_CHECK();
    const double hs[1] = {(double)sweeps + (hrot != 0 ? 0.5 : 0.0)};  // x.5 = sweep cap hit before convergence
    MVF_CHECK_HIP(hipMemcpyAsync(einfo, hs, sizeof(double), hipMemcpyHostToDevice, st));
    MVF_CHECK_HIP(rb_sync(st));
    return 0;
}

extern "C" size_t mvf_solve_minnorm_lr_workspace_bytes(int64_t m, int nrhs) {
    (void)nrhs;
    if (m <= 0) return 0;
    return lr_plan(m).total;
}

static int lr_solve(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond, const double* R,
                    int64_t m, int nrhs, double* C, int* info, double* einfo, int max_sweeps, int reuse, int rank_hint,
                    void* workspace, size_t workspace_bytes, void* stream, bool deflate, bool allow_direct = true,
                    int form_hint = 0) {
    MVF_REQUIRE(m >= 0 && nrhs >= 1 && nrhs <= 8,
                "mvf_solve_minnorm_lr: need m >= 0 and 1 <= nrhs <= 8 (got m=%lld nrhs=%d)", (long long)m, nrhs);
    MVF_REQUIRE(info && einfo, "mvf_solve_minnorm_lr: null info / einfo");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
        return 0;
    }
    MVF_REQUIRE(G && K && R && C, "mvf_solve_minnorm_lr: null pointer");
    MVF_REQUIRE(std::isfinite(lambda_sigma2) && lambda_sigma2 >= 0.0 && tolf > 0.0 && tolf <= 1.0 && rcond >= 0.0,
                "mvf_solve_minnorm_lr: bad regularisation / tolerance factor / rcond");
    MVF_REQUIRE(m <= 65535 - 64, "mvf_solve_minnorm_lr: m too large (%lld)", (long long)m);
    if (max_sweeps <= 0) max_sweeps = 60;
    const LrPlan p = lr_plan(m);
    MVF_REQUIRE(workspace && workspace_bytes >= (deflate ? p.total_d : p.total),
                "mvf_solve_minnorm_lr: workspace too small (%zu < %zu)", workspace_bytes, deflate ? p.total_d : p.total);
    char* ws = (char*)workspace;
    double* S = (double*)(ws + p.off_s);
    double* Y = (double*)(ws + p.off_y);
    double* dg = (double*)(ws + p.off_dg);
    double* pm = (double*)(ws + p.off_pm);
    double* xv = (double*)(ws + p.off_x);
    PcholState* stt = (PcholState*)(ws + p.off_state);
    int* order = (int*)(ws + p.off_order);
    double* piv = (double*)(ws + p.off_piv);
    double* Spart = (double*)(ws + p.off_spart);
    double* Jbuf = (double*)(ws + p.off_j);
    int* flags = (int*)(ws + p.off_flags);
    int* mod = (int*)(ws + p.off_stamps);
    double* sig2 = (double*)(ws + p.off_sig2);
    double* T = (double*)(ws + p.off_t);
    double* part = (double*)(ws + p.off_part);
    unsigned int* rot = (unsigned int*)(ws + p.off_rot);
    double* scal = (double*)(ws + p.off_scal);
    const int64_t mp = p.mp;
    PcholState hs;
    const bool timing = debug_opt(DBG_LR_TIMING) != 0;  // developer option: phase times on stderr
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // the timing events (developer option) are destroyed on EVERY path out of this function, early error returns included
    // (they leaked there); an explicit drop_events() before a tail call keeps them from outliving their use
    auto drop_events = [&ev]() {
        for (auto& e : ev)
            if (e) {
                (void)hipEventDestroy(e);
                e = nullptr;
            }
    };
    struct EventGuard {
        decltype(drop_events)& drop;
        ~EventGuard() { drop(); }
    } event_guard{drop_events};
    if (timing && !reuse) {
        for (auto& e : ev) MVF_CHECK_HIP(hipEventCreate(&e));
        MVF_CHECK_HIP(hipEventRecord(ev[0], st));
    }

    // the truncated solve from the orthogonalised factor (rows of Y = sigma_i u_i^T): C = Y^T (g .* (Y R))
    auto backsolve = [&](int64_t rp, double* ei) -> int {
        const int bsplit = (int)std::min<int64_t>(16, rp / 64);
        const int rows_per_split = (int)cdiv(rp, bsplit);
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(rp, 4)), dim3(256), 0, st, Y, rp, mp, m, R, nrhs, sig2, T);
        hipLaunchKernelGGL(jac_scale_kernel, dim3(1), dim3(256), 0, st, sig2, rp, scal, rcond, T, ei);
        hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)cdiv(mp, 64), (unsigned)bsplit), dim3(256), 0, st, Y, rp, mp, m,
                           T, rows_per_split, part);
        hipLaunchKernelGGL(jac_back_reduce_kernel, dim3((unsigned)cdiv(m * nrhs, 256)), dim3(256), 0, st, part, bsplit, m,
                           nrhs, C);
        MVF_LAUNCH_CHECK();
        return 0;
    };

    // the deflated truncated solve from (Y, Minv in S, the deflation vectors): C = Y^T Pc Minv Pc Minv Pc (Y R)
    auto defl_apply = [&](int64_t rp, int b) -> int {
        const DeflBuf d = defl_layout(rp);
        char* dw = ws + p.off_d;
        double *Ta = (double*)(dw + d.ta), *Tb = (double*)(dw + d.tb), *cb = (double*)(dw + d.cb);
        double *dummy = (double*)(dw + d.dummy), *dpart = (double*)(dw + d.part), *Wsel = (double*)(dw + d.wsel);
        const double* Minv = S;
        auto project = [&](double* T) {  // T -= Wsel^T (Wsel T)
            // (a fused single-workgroup form of this projection for 64-row blocks measured 63 us per call against ~10 for these
            // three launches - one compute unit pulling 256 KB through dependent loads: profiles/r06_c2_step_timeline_b.md)
            hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, Wsel, (int64_t)b, rp, rp, T, 8,
                               dummy, cb);
            hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)(rp / 64), 4u), dim3(256), 0, st, Wsel, (int64_t)b, rp, rp, cb,
                               b / 4, dpart);
            hipLaunchKernelGGL(defl_sub_kernel, dim3((unsigned)cdiv(rp * 8, 256)), dim3(256), 0, st, dpart, 4, rp, T);
        };
        auto apply_inv = [&](const double* Tin, double* Tout) {  // Tout = Minv Tin (Minv symmetric: rows dot Tin)
            hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(rp, 4)), dim3(256), 0, st, Minv, rp, rp, rp, Tin, 8,
                               dummy, Tout);
        };
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(rp, 4)), dim3(256), 0, st, Y, rp, mp, m, R, nrhs, dummy, Ta);
        project(Ta);
        apply_inv(Ta, Tb);
        project(Tb);
        apply_inv(Tb, Ta);
        project(Ta);
        const int bsplit = (int)std::min<int64_t>(16, rp / 64);
        const int rows_per_split = (int)cdiv(rp, bsplit);
        hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)cdiv(mp, 64), (unsigned)bsplit), dim3(256), 0, st, Y, rp, mp, m,
                           Ta, rows_per_split, part);
        hipLaunchKernelGGL(jac_back_reduce_kernel, dim3((unsigned)cdiv(m * nrhs, 256)), dim3(256), 0, st, part, bsplit, m,
                           nrhs, C);
        MVF_LAUNCH_CHECK();
        return 0;
    };

    // the direct form's solve from (E = Rc^-T and its transpose, the deflation vectors, the pivot order): C = Pi^T Pc E E^T Pc Pi R
    int* dflag = (int*)(ws + p.off_dflag);
    auto direct_apply = [&](int64_t rp, int b) -> int {
        const DeflBuf d = defl_layout(rp);
        char* dw = ws + p.off_d;
        double *Ta = (double*)(dw + d.ta), *Tb = (double*)(dw + d.tb), *cb = (double*)(dw + d.cb);
        double *dummy = (double*)(dw + d.dummy), *dpart = (double*)(dw + d.part), *Wsel = (double*)(dw + d.wsel);
        auto project = [&](double* T) {  // T -= Wsel^T (Wsel T)
            hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, Wsel, (int64_t)b, rp, rp, T, 8,
                               dummy, cb);
            hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)(rp / 64), 4u), dim3(256), 0, st, Wsel, (int64_t)b, rp, rp, cb,
                               b / 4, dpart);
            hipLaunchKernelGGL(defl_sub_kernel, dim3((unsigned)cdiv(rp * 8, 256)), dim3(256), 0, st, dpart, 4, rp, T);
        };
        hipLaunchKernelGGL(perm_rows_kernel, dim3((unsigned)cdiv(rp * 8, 256)), dim3(256), 0, st, R, nrhs, order, m, rp, dflag, Ta);
        project(Ta);
        // E^T (Pc b), then E (...): rows of Et / E dot the 8-column block (E = Rc^-T sits behind the factor in the
        // factorisation's workspace, Et in the slot of the permuted matrix)
        const double* E = (const double*)(dw + d.cw) + rp * rp;
        const double* Et = (const double*)(dw + d.s2);
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(rp, 4)), dim3(256), 0, st, Et, rp, rp, rp, Ta, 8, dummy, Tb);
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(rp, 4)), dim3(256), 0, st, E, rp, rp, rp, Tb, 8, dummy, Ta);
        project(Ta);
        hipLaunchKernelGGL(unperm_rows_kernel, dim3((unsigned)cdiv(m * 8, 256)), dim3(256), 0, st, Ta, nrhs, order, m, dflag, C);
        MVF_LAUNCH_CHECK();
        return 0;
    };

    if (reuse) {
        // the workspace still holds the decomposition of the previous call for this matrix
        MVF_CHECK_HIP(rb_copy(st, &hs, stt, sizeof(hs)));
        MVF_CHECK_HIP(rb_sync(st));
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
        if (hs.r <= 0) {
            MVF_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)m * nrhs * sizeof(double), st));
            return 0;
        }
        if (hs.defl) {
            MVF_REQUIRE(workspace_bytes >= p.total_d, "mvf_solve_minnorm_lr: reuse of a deflated decomposition needs its workspace");
            MVF_REQUIRE(hs.defl_block == DEFL_B || hs.defl_block == DEFL_B / 2 || hs.defl_block == DEFL_TINY,
                        "mvf_solve_minnorm_lr: corrupt deflation state");
            if (hs.defl == 2) {  // the direct form answered the previous call (dflag is still 1 from it)
                MVF_REQUIRE(hs.r == m, "mvf_solve_minnorm_lr: corrupt direct-form state");
                return direct_apply(cdiv(hs.r, 64) * 64, hs.defl_block);
            }
            return defl_apply(cdiv(hs.r, 64) * 64, hs.defl_block);
        }
        return backsolve(cdiv(hs.r, 64) * 64, einfo + 6);
    }

    // 1. S = G + ls2 K (zero padding), lambda_max estimate, pivoted Cholesky -> rows 0 .. r-1 of Y
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mp), dim3(256), 0, st, G, K, lambda_sigma2,
                       m, mp, S, scal);
    // lambda_max: power iteration.  With a rank hint it starts from the dominant eigenvector the previous call on this
    // workspace (the previous EM iteration: a nearby matrix) ended with - the Rayleigh quotient is then converged to ~1e-12
    // after 5 steps where the cold start needs 12 for 1e-8 (and, at M = 500, a second run on S2: 0.45 ms of a 3 ms solve).
    // A workspace that does not hold a finished call falls back to the cold start vector on the device; the convergence
    // test before the deflated solve then refines as before.
    double* xkeep = (double*)(ws + p.off_xkeep);
    const int use_hint0 = rank_hint > 0 && rank_hint <= m;
    // (form_hint bit 4, asynchronous entry only: the CALLER knows that the matrix moved since the previous call - sigma^2
    // changed by more than a few per cent, the first iterations of a fit - and asks for the 13 steps the synchronous entry
    // adds after reading the quotient; without it a fit's early asynchronous attempts fail the lambda_max test and cost a
    // repeat plus a four-call cool-down)
    const bool more_power = (form_hint & 4) != 0;
    form_hint &= 3;
    const int npow = use_hint0 ? (more_power ? 13 : 5) : (deflate ? 12 : 8);  // the deflated solve takes its cut-off from this estimate
    hipLaunchKernelGGL(lr_power_kernel, dim3(1), dim3(256), 0, st, xv, xv + mp, m, mp, use_hint0 ? 2 : 1, stt,
                       (const double*)xkeep, rank_hint, (double*)nullptr);
    for (int it = 0; it < npow; ++it) {
        hipLaunchKernelGGL(lr_symv_kernel, dim3((unsigned)cdiv(mp, 4)), dim3(256), 0, st, S, mp, xv, xv + mp);
        hipLaunchKernelGGL(lr_power_kernel, dim3(1), dim3(256), 0, st, xv, xv + mp, m, mp, 0, stt, (const double*)nullptr, 0,
                           it == npow - 1 ? xkeep : (double*)nullptr);
    }
    // 1'. the direct form (see direct_prepare_kernel): the previous call on this workspace kept all m columns
    const int64_t rpm = cdiv(m, 64) * 64;
    if (deflate && allow_direct && use_hint0 && rank_hint == m && m >= 2 * DEFL_TINY && m <= 640 && 2 * rpm <= 65535 &&
        debug_opt(DBG_LR_NO_DEFLATE) == 0 && debug_opt(DBG_LR_NO_DIRECT) == 0) {
        const int64_t rp = rpm;
        const int b = DEFL_TINY;
        const DeflBuf d = defl_layout(rp);
        char* dw = ws + p.off_d;
        double *Ap = (double*)(dw + d.s2), *Za = (double*)(dw + d.za), *Zb = (double*)(dw + d.zb);
        double *Wsel = (double*)(dw + d.wsel), *Gb = (double*)(dw + d.g), *H = (double*)(dw + d.h), *Yh = (double*)(dw + d.yh);
        double *theta = (double*)(dw + d.theta), *dummy = (double*)(dw + d.dummy);
        double* Minv = S;
        // what the previous call left (64 bytes; the stream is idle behind the power iteration's few launches by now): a
        // workspace without a finished factorisation of all m columns skips this form at once, and a previous DIRECT call
        // lets the block iteration continue from its converged block
        const bool nosync = form_hint != 0;  // mvf_solve_minnorm_lrd_async: no status read anywhere in this form
        bool prev_direct = form_hint == 2 && debug_opt(DBG_DEFL_APPS) == 0;
        if (!nosync) {
            PcholState hprev;
            MVF_CHECK_HIP(rb_copy(st, &hprev, stt, sizeof(hprev)));
            MVF_CHECK_HIP(rb_sync(st));
            if (!(hprev.magic == PCHOL_MAGIC && hprev.order_len == (int)m)) {
                if (timing)
                    drop_events();
                return lr_solve(G, K, lambda_sigma2, tolf, rcond, R, m, nrhs, C, info, einfo, max_sweeps, reuse, rank_hint, workspace,
                                workspace_bytes, stream, deflate, false);
            }
            if (hprev.direct_skip > 0 && hprev.direct_skip <= 4) {  // a recent attempt of this form failed: not again just yet
                const int left[1] = {hprev.direct_skip - 1};
                MVF_CHECK_HIP(hipMemcpyAsync(&stt->direct_skip, left, sizeof(left), hipMemcpyHostToDevice, st));
                MVF_CHECK_HIP(rb_sync(st));
                if (timing)
                    drop_events();
                return lr_solve(G, K, lambda_sigma2, tolf, rcond, R, m, nrhs, C, info, einfo, max_sweeps, reuse, rank_hint, workspace,
                                workspace_bytes, stream, deflate, false);
            }
            prev_direct = hprev.defl == 2 && hprev.pad4 == 1 && hprev.defl_block == b &&
                                     debug_opt(DBG_DEFL_APPS) == 0;  // (defl_apps set: the cold three-application plan, for A/B)
            // (the same copy carries THIS call's power iteration: early in a fit the matrix still moves a lot between two calls -
            // sigma^2 falls by an order of magnitude - and five warm steps may not have settled the Rayleigh quotient; eight more,
            // 13 in all as on the cold path, cost 50 us where a failed attempt of this form costs 1.9 ms)
            if (!(std::fabs(hprev.lmax_est - hprev.lmax_prev) <= 1e-7 * hprev.lmax_est)) {
                for (int it = 0; it < 8; ++it) {
                    hipLaunchKernelGGL(lr_symv_kernel, dim3((unsigned)cdiv(mp, 4)), dim3(256), 0, st, S, mp, xv, xv + mp);
                    hipLaunchKernelGGL(lr_power_kernel, dim3(1), dim3(256), 0, st, xv, xv + mp, m, mp, 0, stt, (const double*)nullptr, 0,
                                       it == 7 ? xkeep : (double*)nullptr);
                }
            }
        }
        hipLaunchKernelGGL(direct_prepare_kernel, dim3(1), dim3(256), 0, st, S, m, mp, tolf, stt, info, dflag, form_hint);
        MVF_LAUNCH_CHECK();
        CholPlan cs, cq;
        // (the permuted matrix A[order][order] is gathered straight into the factorisation's work matrix)
        if (int rc = chol_factor_mat_inv(st, S, mp, m, dw + d.cw, &cs, info, 1, order, dflag)) return rc;
        hipLaunchKernelGGL(direct_check_kernel, dim3(1), dim3(256), 0, st, cs.rdiag, m, stt, info, piv, dflag);
        const double* E = cs.W + rp * rp;                           // Rc^-T (upper triangular, identity on the padding)
        (void)Minv;
        auto orthonormalise = [&](const double* Zin, double* Zout) -> int {  // Cholesky QR on the rows
            gemm<false, true>(st, Zin, rp, Zin, rp, Gb, b, b, b, rp);
            if (int rc = chol_factor_mat_inv(st, Gb, b, b, dw + d.cwb, &cq, info, 1)) return rc;
            gemm<true, false>(st, cq.W + (size_t)b * b, b, Zin, rp, Zout, rp, b, rp, b);
            return 0;
        };
        // block inverse iteration from the unit vectors of the b smallest pivots, three applications of A_perm^-1 = E E^T -
        // applied factor by factor (Z E, then (Z E) E^T): the product E E^T is never formed
        // When the previous call on this workspace was answered by this very form, its converged block is still in Za (same
        // pivot order, a nearby matrix): ONE application from there replaces the three from the unit vectors.
        if (!prev_direct) {
            gemm<false, true>(st, E + (m - b) * rp, rp, E, rp, Zb, rp, b, rp, rp);  // rows m-b .. m-1 of E E^T
            if (int rc = orthonormalise(Zb, Za)) return rc;
        }
        for (int ap = prev_direct ? 2 : 1; ap < 3; ++ap) {
            gemm<false, false>(st, Za, rp, E, rp, Wsel, rp, b, rp, rp);
            gemm<false, true>(st, Wsel, rp, E, rp, Zb, rp, b, rp, rp);
            if (int rc = orthonormalise(Zb, Za)) return rc;
        }
        // Rayleigh-Ritz: H = Za A_perm Za^T as the Gram matrix of B = Za Rc (the graded factor, not the assembled matrix), one
        // 64 x 64 Jacobi tile diagonalised in a single launch.  Lo = tril(Rc) goes to the slot of the factor rows (unused in
        // this form), E^T to the slot of the permuted matrix (used up by the factorisation).
        double* Lo = Y;
        hipLaunchKernelGGL(tril_and_transpose_kernel, dim3((unsigned)cdiv(rp, 256), (unsigned)rp), dim3(256), 0, st, cs.W, E, rp, Lo,
                           Ap);
        gemm<false, false>(st, Za, rp, Lo, rp, Zb, rp, b, rp, rp);
        gemm<false, true>(st, Zb, rp, Zb, rp, H, b, b, b, rp);
        if (int rc = chol_factor_mat_inv(st, H, b, b, dw + d.cwb, &cq, info, 0)) return rc;
        const int hnb = b / JB;  // 2: one pair
        hipLaunchKernelGGL(jac_init_kernel, dim3(1u, 1u), dim3(256), 0, st, cq.W, (int64_t)b, (int64_t)b, Yh, mod,
                           hnb + hnb * hnb, rot, 64);  // (+ the stamps, the rotation counter and the diagnostics behind it)
        const double htol = RR_TOL;
        int* hclean = mod + hnb;
        int hsweeps = 0;
        unsigned int hrot2 = 1;
        while (hsweeps < std::max(2, max_sweeps / 12)) {  // each launch runs up to 12 sweeps
            if (hsweeps > 0) MVF_CHECK_HIP(hipMemsetAsync(rot, 0, 256, st));
            hipLaunchKernelGGL(jac_gram_kernel, dim3(1u, 1u), dim3(256), 0, st, Yh, (int64_t)b, hnb, 0, 1, 1, mod, hclean, Spart);
            // (the first launch starts from the previous call's total rotation when that call was a direct one; a second
            // launch - never seen - would continue from the rotated factor and must start cold)
            hipLaunchKernelGGL(jac_eig_kernel<true>, dim3(1u), dim3(EIG_THREADS), 0, st, Spart, 1, htol, hnb, 0, 1 + hsweeps, mod,
                               hclean, Jbuf, flags, rot, 12, hsweeps == 0 ? (const int*)(dflag + 1) : (const int*)nullptr,
                               hsweeps == 0 ? (double*)(ws + p.off_jkeep) : (double*)nullptr);
            hipLaunchKernelGGL(jac_update_kernel, dim3(1u, 1u), dim3(256), 0, st, Yh, (int64_t)b, hnb, 0, Jbuf, flags);
            MVF_LAUNCH_CHECK();
            ++hsweeps;
            if (nosync) break;  // (its convergence - rot[0] == 0 - is part of the device-side acceptance test)
            MVF_CHECK_HIP(rb_copy(st, &hrot2, rot, sizeof(unsigned int)));
            MVF_CHECK_HIP(rb_sync(st));
            if (hrot2 == 0) break;
        }
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, Yh, (int64_t)b, (int64_t)b, (int64_t)b,
                           R, 0, theta, dummy);
        hipLaunchKernelGGL(defl_select_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, theta, b, stt, rcond, m, Yh, einfo);
        gemm<false, false>(st, Yh, b, Za, rp, Wsel, rp, b, rp, b);  // rows = the Ritz vectors to deflate (zero rows else)
        MVF_LAUNCH_CHECK();
        if (int rc = direct_apply(rp, b)) return rc;
        if (nosync) {
            const long long dacc0 = debug_opt(DBG_DIRECT_ACCEPT);
            hipLaunchKernelGGL(direct_close_kernel, dim3(1), dim3(1), 0, st, stt, einfo, info, dflag, rot, (int)m, b,
                               dacc0 > 0 ? (double)(dacc0 - 1) : (double)DEFL_TINY_ACCEPT);
            MVF_LAUNCH_CHECK();
            if (timing)
                drop_events();
            return 0;
        }
        // ONE device -> host copy decides (einfo[0..5], info, the state flag, the power iteration's last two quotients), and
        // ONE launch writes the closing state + einfo: seven separate small copies used to cost 0.13 ms of this 1.5 ms call
        double* rep = (double*)(ws + p.off_scal) + 16;  // 11 doubles of the 256-byte scalar slot (its first two are the shift's)
        hipLaunchKernelGGL(direct_report_kernel, dim3(1), dim3(64), 0, st, einfo, info, dflag, stt, rep);
        double hrep[11];
        MVF_CHECK_HIP(rb_copy(st, hrep, rep, sizeof(hrep)));
        MVF_CHECK_HIP(rb_sync(st));
        const double* he = hrep;
        const int hinfo2 = (int)hrep[6], hflag = (int)hrep[7];
        const double q_est = hrep[8], q_prev = hrep[9], q_maxdiag = hrep[10];
        // the cut-off needs a converged lambda_max: the warm power iteration's last two Rayleigh quotients must agree
        const bool lmax_ok = std::fabs(q_est - q_prev) <= 1e-7 * q_est || q_est == q_maxdiag;
        // (developer option direct_accept = v > 0: accept at most v - 1 deflated directions - forces the fall-back in tests)
        const long long dacc = debug_opt(DBG_DIRECT_ACCEPT);
        const double accept_n = dacc > 0 ? (double)(dacc - 1) : (double)DEFL_TINY_ACCEPT;
        const bool ok = hinfo2 == 0 && hflag == 1 && hrot2 == 0 && lmax_ok && he[4] <= accept_n && std::isfinite(he[5]) &&
                        he[5] > 0.0;
        if (timing) {
            MVF_CHECK_HIP(hipEventRecord(ev[3], st));
            MVF_CHECK_HIP(hipEventSynchronize(ev[3]));
            float t03 = 0;
            (void)hipEventElapsedTime(&t03, ev[0], ev[3]);
            fprintf(stderr, "[mvf_solve_minnorm_lrd] m %lld direct form: %.2f ms (%s block, %d launch(es) of the 64 x 64 Rayleigh-Ritz, %d below the cut, info %d, state %d, lambda_max %s)%s\n",
                    (long long)m, t03, prev_direct ? "continued" : "fresh", hsweeps, (int)he[4], hinfo2, hflag,
                    lmax_ok ? "converged" : "NOT converged", ok ? "" : " -> factor form");
            unsigned int hsw[24];
            (void)hipMemcpy(hsw, rot + 8, sizeof(hsw), hipMemcpyDeviceToHost);
            fprintf(stderr, "    Rayleigh-Ritz sweeps (active rounds / rotations):");
            for (int q = 0; q < 12; ++q) fprintf(stderr, " %u/%u", hsw[2 * q], hsw[2 * q + 1]);
            fprintf(stderr, "\n");
            drop_events();
        }
        if (ok) {
            // the workspace now holds: the same pivot order (all m columns), their pivots, E = Rc^-T and its transpose, the block
            // and the deflation vectors; stream-ordered, no further synchronisation (the caller's next copy sees it)
            hipLaunchKernelGGL(direct_finish_kernel, dim3(1), dim3(1), 0, st, stt, einfo, (int)m, b, hsweeps);
            MVF_LAUNCH_CHECK();
            return 0;
        }
        // anything else: the factor form answers (re-assembles S, which now holds the inverse), and the next four calls on this
        // workspace go to it directly - a system on which this form keeps failing must not pay for the attempt every time
        {
            const int skip[1] = {4};
            MVF_CHECK_HIP(hipMemcpyAsync(&stt->direct_skip, skip, sizeof(skip), hipMemcpyHostToDevice, st));
            MVF_CHECK_HIP(rb_sync(st));
        }
        return lr_solve(G, K, lambda_sigma2, tolf, rcond, R, m, nrhs, C, info, einfo, max_sweeps, reuse, rank_hint, workspace,
                        workspace_bytes, stream, deflate, false);
    }

    int* hint = (int*)(ws + p.off_hint);
    double* Lcc = (double*)(ws + p.off_lcc);
    int* candb = (int*)(ws + p.off_cand);
    const int msteps = (int)m;  // every row of Y retires one column
    const dim3 ugrid((unsigned)(mp / 64), (unsigned)(mp / 64));
    int j = 0, cur = 0, hinfo = 0;
    const int use_hint = rank_hint > 0 && rank_hint <= m;
    // with rank_hint > 0 the factorisation first follows the pivot order the previous call left in this workspace
    if (use_hint) MVF_CHECK_HIP(hipMemcpyAsync(hint, order, (size_t)mp * sizeof(int), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(pchol_init_kernel, dim3(1), dim3(256), 0, st, S, m, mp, tolf, dg, pm, p.nwg, stt, info, use_hint,
                       rank_hint);
    MVF_LAUNCH_CHECK();
    bool finished = false;
    if (use_hint) {
        const int npan = (int)cdiv(rank_hint, 64);
        for (int b = 0; b < npan; ++b) {
            hipLaunchKernelGGL(pchol_panel_factor_kernel, dim3(1), dim3(256), 0, st, S, m, mp, hint, rank_hint, b, dg, stt,
                               order, piv, Lcc, candb);
            hipLaunchKernelGGL(pchol_panel_rows_kernel, dim3((unsigned)p.nwg), dim3(4 * PC_T), 0, st, S, Y, mp, b, dg, pm, stt,
                               Lcc, candb);
            hipLaunchKernelGGL(pchol_update_kernel, ugrid, dim3(256), 0, st, S, Y, mp, 64 * b, stt, b + 1);
        }
        MVF_LAUNCH_CHECK();
        MVF_CHECK_HIP(rb_copy(st, &hs, stt, sizeof(hs)));
        MVF_CHECK_HIP(rb_copy(st, &hinfo, info, sizeof(int)));
        MVF_CHECK_HIP(rb_sync(st));
        if (hinfo != 0) return 0;
        j = hs.r;  // pivots taken from the hint (the last accepted panel may be partial: its rows behind are zero and
                   // already applied to S, the greedy steps overwrite them)
        finished = hs.done || j >= msteps;
    }
    int jbase = j;  // first row not yet applied to S by a trailing update
    // up to 512 columns: the steps of a 32-row sub-block in ONE single-workgroup launch (pchol_steps_kernel), the whole rest of the
    // factorisation enqueued at once - launches behind the end of the factorisation return at their first instruction.  (A device
    // that refuses 129 KB of dynamic LDS keeps the one-launch-per-step form.)
    const bool one_wg = mp <= PS_T && hipFuncSetAttribute(reinterpret_cast<const void*>(pchol_steps_kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)PS_LDS) == hipSuccess;
    if (!one_wg) (void)hipGetLastError();
    auto enqueue = [&](int upto) {
        while (j < upto) {
            if (one_wg) {
                const int sb = jbase + ((j - jbase) / PS_SUB) * PS_SUB;
                const int n = std::min(upto - j, sb + PS_SUB - j);
                hipLaunchKernelGGL(pchol_steps_kernel, dim3(1), dim3(PS_T), PS_LDS, st, S, Y, mp, jbase, sb, j, n,
                                   dg + (size_t)cur * mp, stt, order, piv);
                j += n;
            } else {
                hipLaunchKernelGGL(pchol_step_kernel, dim3((unsigned)p.nwg), dim3(PC_T), 0, st, S, Y, mp, jbase, j,
                                   dg + (size_t)cur * mp, dg + (size_t)(cur ^ 1) * mp, pm + (size_t)cur * 2 * p.nwg,
                                   pm + (size_t)(cur ^ 1) * 2 * p.nwg, p.nwg, stt, order, piv);
                cur ^= 1;
                ++j;
            }
            if (j - jbase == 64 && j < msteps) {
                hipLaunchKernelGGL(pchol_update_kernel, ugrid, dim3(256), 0, st, S, Y, mp, jbase, stt, -1);
                jbase = j;
            }
        }
    };
    // without a hint: the first 256 steps, then 128 more per status read; after hint panels only the tail is left: 32
    // steps per status read
    const bool tail_only = use_hint && j > 0;
    // (behind the hint panels the previous call's rank says how long the greedy tail will be: every enqueued step costs the
    // host ~8 us whether it still has a pivot to take or not - a fixed batch of 32 was 250 us of launches for ~10 pivots)
    const int tail_first = std::max(8, std::min(32, rank_hint - j + 8));
    int upto = one_wg ? msteps : (tail_only ? std::min(msteps, j + tail_first) : std::min(msteps, 256));
    while (!finished) {
        enqueue(upto);
        MVF_LAUNCH_CHECK();
        MVF_CHECK_HIP(rb_copy(st, &hs, stt, sizeof(hs)));
        MVF_CHECK_HIP(rb_copy(st, &hinfo, info, sizeof(int)));
        MVF_CHECK_HIP(rb_sync(st));
        if (hinfo != 0) return 0;  // non-finite input: info[0] tells the caller
        if (hs.done || j >= msteps) break;
        upto = std::min(msteps, upto + (tail_only ? 16 : 128));
    }
    // this workspace now holds a finished order of hs.r rows (and the dominant eigenvector of this call: keep_len = m)
    const int tag[6] = {PCHOL_MAGIC, (int)hs.r, (int)m, 0, 0, 0};
    MVF_CHECK_HIP(hipMemcpyAsync(&stt->magic, tag, sizeof(tag), hipMemcpyHostToDevice, st));  // (tag lives until the
                                                                                                // final synchronise)
    const int64_t r = hs.r;
    if (timing) MVF_CHECK_HIP(hipEventRecord(ev[1], st));
    const double hr[1] = {(double)r};
    MVF_CHECK_HIP(hipMemcpyAsync(einfo + 6, hr, sizeof(double), hipMemcpyHostToDevice, st));
    if (r == 0) {  // the zero matrix: minimum-norm solution 0
        MVF_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)m * nrhs * sizeof(double), st));
        MVF_CHECK_HIP(hipMemsetAsync(einfo, 0, 6 * sizeof(double), st));
        if (deflate) MVF_CHECK_HIP(hipMemsetAsync(einfo + 7, 0, 3 * sizeof(double), st));  // no block ran
        MVF_CHECK_HIP(rb_sync(st));
        return 0;
    }
    const int64_t rp = cdiv(r, 64) * 64;
    if (rp > r) MVF_CHECK_HIP(hipMemsetAsync(Y + r * mp, 0, (size_t)(rp - r) * mp * sizeof(double), st));

    // 2'. deflated solve: only the invariant subspace below the cut-off is computed (see DEFL_B above)
    // (2 rp rows of the factor-with-inverse layout must fit a launch grid: larger factors take the Jacobi path)
    if (deflate && r >= 2 * DEFL_TINY && 2 * rp <= 65535 && debug_opt(DBG_LR_NO_DEFLATE) == 0) {
        const DeflBuf d = defl_layout(rp);
        char* dw = ws + p.off_d;
        double *S2 = (double*)(dw + d.s2), *Za = (double*)(dw + d.za), *Zb = (double*)(dw + d.zb);
        double *Wsel = (double*)(dw + d.wsel), *Gb = (double*)(dw + d.g), *H = (double*)(dw + d.h), *Yh = (double*)(dw + d.yh);
        double *theta = (double*)(dw + d.theta), *dummy = (double*)(dw + d.dummy);
        double* Minv = S;  // the assembled matrix is used up: S now holds S2^-1 (rp x rp)
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
        gemm<false, true>(st, Y, mp, Y, mp, S2, rp, rp, rp, mp, 1);  // S2 = L^T L (identity-free zero padding)
        // the cut-off is rcond x the power iteration's Rayleigh quotient: converged after the 12 steps when lambda_2 /
        // lambda_1 <~ 0.5 (kernel Gram matrices: 0.45); a clustered top of the spectrum gets more steps, on S2 (same
        // non-zero eigenvalues as L L^T), until two successive quotients agree to 1e-7 or 512 steps are spent
        if (!(std::fabs(hs.lmax_est - hs.lmax_prev) <= 1e-7 * hs.lmax_est)) {
            const double before = hs.lmax_est;
            PcholState h2 = hs;
            hipLaunchKernelGGL(lr_power_kernel, dim3(1), dim3(256), 0, st, xv, xv + mp, r, rp, 1, stt);
            for (int batch = 0; batch < 32; ++batch) {
                for (int it = 0; it < 16; ++it) {
                    hipLaunchKernelGGL(lr_symv_kernel, dim3((unsigned)cdiv(rp, 4)), dim3(256), 0, st, S2, rp, xv, xv + mp);
                    hipLaunchKernelGGL(lr_power_kernel, dim3(1), dim3(256), 0, st, xv, xv + mp, r, rp, 0, stt);
                }
                MVF_LAUNCH_CHECK();
                MVF_CHECK_HIP(rb_copy(st, &h2, stt, sizeof(h2)));
                MVF_CHECK_HIP(rb_sync(st));
                if (std::fabs(h2.lmax_est - h2.lmax_prev) <= 1e-7 * h2.lmax_est) break;
            }
            const double best[1] = {std::max(before, h2.lmax_est)};  // both are lower bounds of lambda_max
            MVF_CHECK_HIP(hipMemcpyAsync(&stt->lmax_est, best, sizeof(best), hipMemcpyHostToDevice, st));
            MVF_CHECK_HIP(rb_sync(st));
        }
        CholPlan cs, cq;
        if (int rc = chol_factor_mat_inv(st, S2, rp, r, dw + d.cw, &cs, info, 1)) return rc;
        const double* E = cs.W + rp * rp;                      // Rc^-T (upper triangular, identity on the padding)
        gemm<false, true>(st, E, rp, E, rp, Minv, rp, rp, rp, rp, 1);  // Minv = Rc^-T Rc^-1
        // Block size: 256 vectors and three applications of Minv (two until round 4); when the previous call on this workspace (the previous EM
        // iteration: rank_hint) truncated at most DEFL_SMALL_MAX directions, 128 vectors and three applications (the
        // 129th eigenvalue is then still > 4 x the cut: tools/lrproto_partial2.py on 60 k x 3000 systems, 65 - 68
        // truncated: field within 5e-6 of the exactly truncated solve).  A 128-block that finds more than DEFL_SMALL_ACCEPT
        // is repeated with 256.
        int hsweeps = 0, b = DEFL_B;
        unsigned int hrot2 = 1;
        double he[6] = {0, 0, 0, 0, 0, 0};
        bool ok = false;
        auto attempt = [&](int napp) -> int {
            auto orthonormalise = [&](const double* Zin, double* Zout) -> int {  // Cholesky QR on the rows
                gemm<false, true>(st, Zin, rp, Zin, rp, Gb, b, b, b, rp);
                if (int rc = chol_factor_mat_inv(st, Gb, b, b, dw + d.cwb, &cq, info, 1)) return rc;
                gemm<true, false>(st, cq.W + (size_t)b * b, b, Zin, rp, Zout, rp, b, rp, b);  // Lg^-1 Zin
                return 0;
            };
            // inverse iteration, started on the unit vectors of the b smallest pivots: Minv e_j = rows r-b .. r-1 of Minv
            // (one Cholesky-QR pass per application leaves the rows orthonormal to ~1e-13: tools/lrproto_partial2.py's
            // pass-count comparison - the Ritz decision and the projector do not need more)
            if (int rc = orthonormalise(Minv + (r - b) * rp, Za)) return rc;
            for (int ap = 1; ap < napp; ++ap) {
                gemm<false, false>(st, Za, rp, Minv, rp, Zb, rp, b, rp, rp);
                if (int rc = orthonormalise(Zb, Za)) return rc;  // Za = the block
            }
            // Rayleigh-Ritz: H = Za S2 Za^T, its eigenvectors by the Jacobi kernels on the Cholesky factor of H
            gemm<false, false>(st, Za, rp, S2, rp, Zb, rp, b, rp, rp);
            gemm<false, true>(st, Zb, rp, Za, rp, H, b, b, b, rp);
            if (int rc = chol_factor_mat_inv(st, H, b, b, dw + d.cwb, &cq, info, 0)) return rc;
            hipLaunchKernelGGL(jac_init_kernel, dim3((unsigned)(b / 64), (unsigned)(b / 64)), dim3(256), 0, st, cq.W, (int64_t)b,
                               (int64_t)b, Yh);
            MVF_LAUNCH_CHECK();
            const int hnb = b / JB, hnp = hnb / 2, hnk = b / 64;
            const double htol = RR_TOL;
            int* hclean = mod + hnb;
            MVF_CHECK_HIP(hipMemsetAsync(mod, 0, (size_t)(hnb + (size_t)hnb * hnb) * sizeof(int), st));
            hsweeps = 0;
            hrot2 = 1;
            // Sweeps are enqueued in batches - ten, then two at a time - with one rotation counter per sweep and ONE status
            // read per batch (round 5 read the counter after every sweep: nine or ten host round trips of ~40 us in a
            // 128-vector Rayleigh-Ritz).  Behind the sweep that converges every pair is clean, so the rest of its batch
            // returns at once (pair_is_clean) and changes nothing: the result and the reported sweep count are those of
            // the one-by-one loop.
            while (hsweeps < max_sweeps && hrot2 != 0) {
                const int batch = hnb == 2 ? 1 : std::min(max_sweeps - hsweeps, hsweeps == 0 ? 10 : 2);  // (9 - 10 at b = 128)
                MVF_CHECK_HIP(hipMemsetAsync(rot, 0, 16 * sizeof(unsigned int), st));
                for (int sb = 0; sb < batch; ++sb)
                    for (int rd = 0; rd < hnb - 1; ++rd) {
                        const int stamp = 1 + (hsweeps + sb) * (hnb - 1) + rd;
                        hipLaunchKernelGGL(jac_gram_kernel, dim3((unsigned)hnp, (unsigned)hnk), dim3(256), 0, st, Yh, (int64_t)b,
                                           hnb, rd, hnk, 1, mod, hclean, Spart);
                        if (rd == 0)  // (a 64-vector block is ONE pair: all its sweeps run inside this launch)
                            hipLaunchKernelGGL(jac_eig_kernel<true>, dim3((unsigned)hnp), dim3(EIG_THREADS), 0, st, Spart, hnk,
                                               htol, hnb, rd, stamp, mod, hclean, Jbuf, flags, rot + sb, hnb == 2 ? 12 : 1);
                        else
                            hipLaunchKernelGGL(jac_eig_kernel<false>, dim3((unsigned)hnp), dim3(EIG_THREADS), 0, st, Spart, hnk,
                                               htol, hnb, rd, stamp, mod, hclean, Jbuf, flags, rot + sb);
                        hipLaunchKernelGGL(jac_update_kernel, dim3((unsigned)hnp, (unsigned)(b / 64)), dim3(256), 0, st, Yh,
                                           (int64_t)b, hnb, rd, Jbuf, flags);
                    }
                MVF_LAUNCH_CHECK();
                unsigned int hb[16];
                MVF_CHECK_HIP(rb_copy(st, hb, rot, sizeof(hb)));
                MVF_CHECK_HIP(rb_sync(st));
                int done_at = -1;
                for (int sb = 0; sb < batch && done_at < 0; ++sb)
                    if (hb[sb] == 0) done_at = sb;
                if (done_at >= 0) {
                    hsweeps += done_at + 1;
                    hrot2 = 0;
                } else {
                    hsweeps += batch;
                }
            }
            hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, Yh, (int64_t)b, (int64_t)b,
                               (int64_t)b, R, 0, theta, dummy);
            hipLaunchKernelGGL(defl_select_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, theta, b, stt, rcond, r, Yh,
                               einfo);
            gemm<false, false>(st, Yh, b, Za, rp, Wsel, rp, b, rp, b);  // rows = the Ritz vectors to deflate (zero rows else)
            MVF_LAUNCH_CHECK();
            return 0;
        };
        // developer options (A/B of the block plan on one system): defl_block = 64 / 128 / 256 forces that block alone,
        // defl_apps the number of applications of S2^-1
        const long long fblock = debug_opt(DBG_DEFL_BLOCK), fapps = debug_opt(DBG_DEFL_APPS);
        const bool forced = (fblock == 64 || fblock == 128 || fblock == 256) && r >= 2 * fblock;
        const bool tiny = r < 2 * DEFL_B;
        const bool small_first = use_hint && hs.defl_nsel > 0 && hs.defl_nsel <= DEFL_SMALL_MAX && r >= 2 * DEFL_B;
        for (int pass = (small_first || tiny || forced) ? 0 : 1; pass < 2 && !ok; ++pass) {
            int napp, accept;
            if (forced) {
                b = (int)fblock;
                napp = 3;
                accept = b - (b == DEFL_TINY ? DEFL_TINY - DEFL_TINY_ACCEPT : DEFL_GUARD);
                pass = 1;  // a single attempt
            } else if (tiny) {
                b = DEFL_TINY;
                napp = 3;
                accept = DEFL_TINY_ACCEPT;
                pass = 1;  // no larger block fits this factor: the Jacobi path answers if this one does not
            } else {
                b = pass == 0 ? DEFL_B / 2 : DEFL_B;
                // three applications for either block since round 5 (the 256-vector block ran two until round 4): one more
                // power of the eigenvalue ratio in the subspace error for 0.3 ms of a 7 ms solve.  What it does NOT do is
                // move the result systematically closer to the reference: seven mathematically equivalent solvers land
                // between 0.88 and 1.19 x the reference floor on the same ten-step fit (profiles/r05_solver_noise.md)
                napp = 3;
                accept = pass == 0 ? DEFL_SMALL_ACCEPT : b - DEFL_GUARD;
            }
            if (fapps >= 1 && fapps <= 8) napp = (int)fapps;
            if (int rc = attempt(napp)) return rc;
            if (timing) MVF_CHECK_HIP(hipEventRecord(ev[2], st));
            if (int rc = defl_apply(rp, b)) return rc;
            MVF_CHECK_HIP(rb_copy(st, he, einfo, sizeof(he)));
            MVF_CHECK_HIP(rb_copy(st, &hinfo, info, sizeof(int)));
            MVF_CHECK_HIP(rb_sync(st));
            ok = hinfo == 0 && hrot2 == 0 && he[4] <= (double)accept && std::isfinite(he[5]) && he[5] > 0.0;
            if (hinfo != 0) break;  // a factorisation met a non-positive pivot: the larger block would meet it too
        }
        if (timing) {
            MVF_CHECK_HIP(hipEventRecord(ev[3], st));
            MVF_CHECK_HIP(hipEventSynchronize(ev[3]));
            float t01 = 0, t12 = 0, t23 = 0;
            (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
            (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
            (void)hipEventElapsedTime(&t23, ev[2], ev[3]);
            fprintf(stderr, "[mvf_solve_minnorm_lrd] m %lld rows %lld: factor %.2f ms, subspace %.2f ms (%d sweeps on %d, %d below the cut, info %d)%s, solve %.2f ms\n",
                    (long long)m, (long long)r, t01, t12, hsweeps, b, (int)he[4], hinfo, ok ? "" : " -> Jacobi path", t23);
        }
        if (ok) {
            const int dtag[4] = {1, b, 0, (int)he[4]};  // defl, defl_block, pad4, defl_nsel (the next call's block choice)
            MVF_CHECK_HIP(hipMemcpyAsync(&stt->defl, dtag, sizeof(dtag), hipMemcpyHostToDevice, st));
            const double hsw[5] = {(double)hsweeps, he[1], he[2], he[3], 0.0};  // [4] = delta = 0 as on the Jacobi path
            MVF_CHECK_HIP(hipMemcpyAsync(einfo, hsw, sizeof(hsw), hipMemcpyHostToDevice, st));
            const double hblk[3] = {(double)b, 1.0, 0.0};  // einfo[7] = the block size used (0: the Jacobi path answered),
            MVF_CHECK_HIP(hipMemcpyAsync(einfo + 7, hblk, sizeof(hblk), hipMemcpyHostToDevice, st));  // [8] = the form, [9] = 0
            MVF_CHECK_HIP(rb_sync(st));
            if (timing)
                drop_events();
            return 0;
        }
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));  // the Jacobi path below starts clean (Y is untouched)
        if (timing) MVF_CHECK_HIP(hipEventRecord(ev[1], st));
    }

    // 2. one-sided block Jacobi on the r rows of Y
    const int nb = (int)(rp / JB), npairs = nb / 2, nk = (int)(mp / 64);
    const int gram_wgs = LR_GRAM_WGS_DEFAULT;
    int nsplit = std::min(nk, std::max(1, gram_wgs / npairs));
    const int kchunks = (int)cdiv(nk, nsplit);
    nsplit = (int)cdiv(nk, kchunks);
    const double tol = std::sqrt((double)m) * 2.220446049250313e-16;
    int* clean = mod + nb;
    MVF_CHECK_HIP(hipMemsetAsync(mod, 0, (size_t)(nb + (size_t)nb * nb) * sizeof(int), st));  // all pairs dirty
    int sweeps = 0;
    unsigned int hrot = 1;
    while (sweeps < max_sweeps && hrot != 0) {  // batches of sweeps, one status read per batch (see the deflated solve above)
        const int batch = std::min(max_sweeps - sweeps, sweeps == 0 ? 8 : 2);
        MVF_CHECK_HIP(hipMemsetAsync(rot, 0, 8 * sizeof(unsigned int), st));
        for (int sb = 0; sb < batch; ++sb)
            for (int rd = 0; rd < nb - 1; ++rd) {
                const int stamp = 1 + (sweeps + sb) * (nb - 1) + rd;
                hipLaunchKernelGGL(jac_gram_kernel, dim3((unsigned)npairs, (unsigned)nsplit), dim3(256), 0, st, Y, mp, nb, rd,
                                   nsplit, kchunks, mod, clean, Spart);
                if (rd == 0)
                    hipLaunchKernelGGL(jac_eig_kernel<true>, dim3((unsigned)npairs), dim3(EIG_THREADS), 0, st, Spart, nsplit,
                                       tol, nb, rd, stamp, mod, clean, Jbuf, flags, rot + sb);
                else
                    hipLaunchKernelGGL(jac_eig_kernel<false>, dim3((unsigned)npairs), dim3(EIG_THREADS), 0, st, Spart, nsplit,
                                       tol, nb, rd, stamp, mod, clean, Jbuf, flags, rot + sb);
                hipLaunchKernelGGL(jac_update_kernel, dim3((unsigned)npairs, (unsigned)(mp / 64)), dim3(256), 0, st, Y, mp, nb,
                                   rd, Jbuf, flags);
            }
        MVF_LAUNCH_CHECK();
        unsigned int hb[8];
        MVF_CHECK_HIP(rb_copy(st, hb, rot, sizeof(hb)));
        MVF_CHECK_HIP(rb_sync(st));
        int done_at = -1;
        for (int sb = 0; sb < batch && done_at < 0; ++sb)
            if (hb[sb] == 0) done_at = sb;
        if (done_at >= 0) {
            sweeps += done_at + 1;
            hrot = 0;
        } else {
            sweeps += batch;
        }
    }

    // 3. truncated minimum-norm solve
    if (timing) MVF_CHECK_HIP(hipEventRecord(ev[2], st));
    if (int rc = backsolve(rp, einfo)) return rc;
    if (timing) {
        MVF_CHECK_HIP(hipEventRecord(ev[3], st));
        MVF_CHECK_HIP(hipEventSynchronize(ev[3]));
        float t01 = 0, t12 = 0, t23 = 0;
        (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
        (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
        (void)hipEventElapsedTime(&t23, ev[2], ev[3]);
        fprintf(stderr, "[mvf_solve_minnorm_lr] m %lld rows %lld (last greedy row %d): factor %.2f ms, jacobi %.2f ms (%d sweeps), solve %.2f ms\n",
                (long long)m, (long long)r, j, t01, t12, sweeps, t23);
#ifdef MVF_EIG_CLOCKS
        unsigned int hc[8];
        (void)hipMemcpy(hc, rot, sizeof(hc), hipMemcpyDeviceToHost);
        fprintf(stderr, "    eig kernel sections (100 MHz ticks): load %u, rounds %u, store %u\n", hc[4], hc[5], hc[6]);
#endif
        drop_events();
    }
    const double hsw[1] = {(double)sweeps + (hrot != 0 ? 0.5 : 0.0)};  // x.5 = sweep cap hit before convergence
    MVF_CHECK_HIP(hipMemcpyAsync(einfo, hsw, sizeof(double), hipMemcpyHostToDevice, st));
    // mvf_solve_minnorm_lrd answered by this path (small factor, launch-grid limit, lr_no_deflate, a failed attempt):
    // einfo[7] = 0, the block size of a deflated solve that did not run (mvf.h)
    if (deflate) MVF_CHECK_HIP(hipMemsetAsync(einfo + 7, 0, 3 * sizeof(double), st));  // block, form, repeat flag
    MVF_CHECK_HIP(rb_sync(st));
    return 0;
}

extern "C" int mvf_solve_minnorm_lr(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond,
                                    const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo,
                                    int max_sweeps, int reuse, int rank_hint, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return lr_solve(G, K, lambda_sigma2, tolf, rcond, R, m, nrhs, C, info, einfo, max_sweeps, reuse, rank_hint, workspace,
                    workspace_bytes, stream, false);
}

extern "C" size_t mvf_solve_minnorm_lrd_workspace_bytes(int64_t m, int nrhs) {
    (void)nrhs;
    if (m <= 0) return 0;
    return lr_plan(m).total_d;
}

extern "C" int mvf_solve_minnorm_lrd(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond,
                                     const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo,
                                     int max_sweeps, int reuse, int rank_hint, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return lr_solve(G, K, lambda_sigma2, tolf, rcond, R, m, nrhs, C, info, einfo, max_sweeps, reuse, rank_hint, workspace,
                    workspace_bytes, stream, true);
}

extern "C" int mvf_solve_minnorm_lrd_async(const double* G, const double* K, double lambda_sigma2, double tolf, double rcond,
                                           const double* R, int64_t m, int nrhs, double* C, int* info, double* einfo,
                                           int form_hint, void* workspace, size_t workspace_bytes, void* stream) {
    MVF_REQUIRE((form_hint & ~4) == 1 || (form_hint & ~4) == 2,
                "mvf_solve_minnorm_lrd_async: form_hint must be 1 or 2, optionally + 4 (got %d)", form_hint);
    MVF_REQUIRE(m >= 2 * DEFL_TINY && m <= 640, "mvf_solve_minnorm_lrd_async: the direct form covers 128 <= m <= 640 (got %lld)",
                (long long)m);
    MVF_REQUIRE(debug_opt(DBG_LR_NO_DEFLATE) == 0 && debug_opt(DBG_LR_NO_DIRECT) == 0,
                "mvf_solve_minnorm_lrd_async: the direct form is switched off (developer option)");
    return lr_solve(G, K, lambda_sigma2, tolf, rcond, R, m, nrhs, C, info, einfo, 60, 0, (int)m, workspace, workspace_bytes,
                    stream, true, true, form_hint);
}

extern "C" int mvf_lr_pivot_order(const void* workspace, size_t workspace_bytes, int64_t m, int* order_out, double* pivots_out,
                                  double* tol_out, int64_t* r_out, void* stream) {
    MVF_REQUIRE(m > 0 && workspace && order_out && r_out, "mvf_lr_pivot_order: bad arguments");
    const LrPlan p = lr_plan(m);
    MVF_REQUIRE(workspace_bytes >= p.total, "mvf_lr_pivot_order: not the workspace of mvf_solve_minnorm_lr for this m");
    hipStream_t st = (hipStream_t)stream;
    const char* ws = (const char*)workspace;
    PcholState hs;
    MVF_CHECK_HIP(rb_copy(st, &hs, ws + p.off_state, sizeof(hs)));
    MVF_CHECK_HIP(rb_sync(st));
    MVF_REQUIRE(hs.magic == PCHOL_MAGIC && hs.r >= 0 && hs.r <= m, "mvf_lr_pivot_order: the workspace holds no finished factorisation");
    MVF_CHECK_HIP(hipMemcpyAsync(order_out, ws + p.off_order, (size_t)hs.r * sizeof(int), hipMemcpyDeviceToHost, st));
    if (pivots_out)
        MVF_CHECK_HIP(hipMemcpyAsync(pivots_out, ws + p.off_piv, (size_t)hs.r * sizeof(double), hipMemcpyDeviceToHost, st));
    MVF_CHECK_HIP(rb_sync(st));
    if (tol_out) *tol_out = hs.tol;
    *r_out = hs.r;
    return 0;
}

extern "C" int mvf_pinv_diag(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, double rcond,
                             int lowrank, double* diag_out, void* workspace, size_t workspace_bytes, mvf_dtype dtype,
                             void* stream) {
    MVF_REQUIRE(n >= 0 && m > 0, "mvf_pinv_diag: need n >= 0 and m > 0");
    MVF_REQUIRE(beta >= 0.0 && std::isfinite(beta) && rcond >= 0.0, "mvf_pinv_diag: bad beta / rcond");
    MVF_REQUIRE(dtype == MVF_F32 || dtype == MVF_F64, "mvf_pinv_diag: bad dtype %d", (int)dtype);
    if (n == 0) return 0;
    MVF_REQUIRE(x4 && ctrl4 && diag_out && workspace, "mvf_pinv_diag: null pointer");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const double *Y, *sig2, *scal;
    double* gw;
    int64_t nrows, mp;
    if (lowrank) {
        const LrPlan p = lr_plan(m);
        MVF_REQUIRE(workspace_bytes >= p.total, "mvf_pinv_diag: not the workspace of mvf_solve_minnorm_lr for this m");
        PcholState hs;
        MVF_CHECK_HIP(rb_copy(st, &hs, ws + p.off_state, sizeof(hs)));
        MVF_CHECK_HIP(rb_sync(st));
        MVF_REQUIRE(hs.magic == PCHOL_MAGIC && hs.r >= 0 && hs.r <= m, "mvf_pinv_diag: the workspace holds no finished decomposition");
        MVF_REQUIRE(!hs.defl, "mvf_pinv_diag: the workspace holds a deflated decomposition (mvf_solve_minnorm_lrd); run mvf_solve_minnorm_lr");
        mp = p.mp;
        nrows = cdiv(hs.r, 64) * 64;
        Y = (const double*)(ws + p.off_y);
        sig2 = (const double*)(ws + p.off_sig2);
        scal = (const double*)(ws + p.off_scal);  // [1] = 0: no shift on this path
        gw = (double*)(ws + p.off_t);
    } else {
        const JacPlan p = jac_plan(m, 1);
        MVF_REQUIRE(workspace_bytes >= p.total, "mvf_pinv_diag: not the workspace of mvf_solve_minnorm for this m");
        CholPlan cq;
        chol_layout(m, 0, workspace, &cq);
        mp = p.mp;
        nrows = p.mp;
        Y = (const double*)(ws + p.off_y);
        sig2 = (const double*)(ws + p.off_sig2);
        scal = cq.scal;  // [1] = the shift delta that the solve subtracted from the eigenvalues
        gw = (double*)(ws + p.off_t);
    }
    if (nrows == 0) {
        MVF_CHECK_HIP(hipMemsetAsync(diag_out, 0, (size_t)n * sizeof(double), st));
        return 0;
    }
    hipLaunchKernelGGL(pinv_weights_kernel, dim3(1), dim3(256), 0, st, sig2, nrows, scal, rcond, gw);
    const double s = std::sqrt(beta * LOG2E);
    const unsigned grid = (unsigned)cdiv(n, 256);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(pinv_diag_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x4, n, (const float*)ctrl4, m,
                           (float)s, Y, nrows, mp, gw, diag_out);
    else
        hipLaunchKernelGGL(pinv_diag_kernel<double>, dim3(grid), dim3(256), 0, st, (const double*)x4, n,
                           (const double*)ctrl4, m, s, Y, nrows, mp, gw, diag_out);
    MVF_LAUNCH_CHECK();
    return 0;
}
