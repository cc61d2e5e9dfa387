// Minimum-norm coefficient solve with the reference's lstsq semantics:
//     C = sum over |lambda_i| > rcond * max|lambda|  of  q_i (q_i^T R) / lambda_i ,   (G + ls2 K) = Q diag(lambda) Q^T
//
// Reference: dynamo `lstsq_solver(lhs, rhs, "scipy")` = scipy.linalg.lstsq = LAPACK gelsd (minimum-norm solution,
// singular values below eps * s_max dropped), as Spateo calls it (spateo/tdr/morphometrics/morphofield/
// sparsevfc.py:110,194,250); in-tree analogue `_pinv(SigmaInv)` spateo/alignment/methods/morpho_class.py:1287.
// lhs is symmetric, so its singular values are |eigenvalues| and the SVD-truncated solution is the formula above.
//
// Algorithm (hand-written for gfx950; no rocSOLVER):
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
#include "mvf_common.h"
#include "mvf_solve.h"

namespace mvf {

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
                                                       double* __restrict__ Y) {
    __shared__ double t[64][65];
    const int bi = blockIdx.y, bj = blockIdx.x;  // tile of L: rows bi*64.., columns bj*64..
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
template <bool FULL>
__global__ __launch_bounds__(EIG_THREADS) void jac_eig_kernel(const double* __restrict__ Spart, int nsplit, double tol,
                                                              int nb, int round, int stamp, int* __restrict__ mod,
                                                              int* __restrict__ clean, double* __restrict__ Jbuf,
                                                              int* __restrict__ flags,
                                                              unsigned int* __restrict__ rot_total) {
    __shared__ double S[JP][JP];
    __shared__ double Jm[JP][JP];
    __shared__ double cs[JB][2];
    const int pair = blockIdx.x, tid = threadIdx.x;
    int bp, bq;
    rr_pair(nb, round, pair, bp, bq);
    if (pair_is_clean(mod, clean, nb, bp, bq)) {  // uniform over the workgroup; nobody writes these stamps this round
        if (tid == 0) flags[pair] = 0;
        return;
    }
    const double* sp = Spart + (int64_t)pair * nsplit * (JP * JP);
    for (int e = tid; e < JP * JP; e += EIG_THREADS) {
        double s = 0.0;
        for (int q = 0; q < nsplit; ++q) s += sp[(int64_t)q * (JP * JP) + e];
        S[e >> 6][e & 63] = s;
        Jm[e >> 6][e & 63] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int l = tid & 31, k = tid >> 5;
    const double tol2 = tol * tol;
    int nrot = 0;
    constexpr int NR = FULL ? JP - 1 : JB;
#pragma unroll 1
    for (int r = 0; r < NR; ++r) {
        int p, q;
        inner_pair<FULL>(r, l, p, q);
        if (k == 0) {
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
            cs[l][0] = c;
            cs[l][1] = sn;
            nrot += __popcll(__ballot(act));  // the 32 lanes of this half-wave = the round's 32 pairs
        }
        __syncthreads();
        const double cl = cs[l][0], sl = cs[l][1], ck = cs[k][0], sk = cs[k][1];
        {
            int pk, qk;
            inner_pair<FULL>(r, k, pk, qk);
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
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = k + 32 * j;
            const double a = Jm[i][p], b = Jm[i][q];
            Jm[i][p] = fma(cl, a, -(sl * b));
            Jm[i][q] = fma(sl, a, cl * b);
        }
        __syncthreads();
    }
    double* jo = Jbuf + (int64_t)pair * (JP * JP);
    for (int e = tid; e < JP * JP; e += EIG_THREADS) jo[e] = Jm[e >> 6][e & 63];
    if (tid == 0) {
        flags[pair] = nrot > 0;
        if (nrot > 0) {
            atomicAdd(rot_total, (unsigned int)nrot);
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
__global__ __launch_bounds__(256) void jac_rowstat_kernel(const double* __restrict__ Y, int64_t mp, int64_t m,
                                                          const double* __restrict__ R, int nrhs,
                                                          double* __restrict__ sig2, double* __restrict__ T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= mp) return;
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
__global__ __launch_bounds__(256) void jac_back_kernel(const double* __restrict__ Y, int64_t mp, int64_t m,
                                                       const double* __restrict__ T, int rows_per_split,
                                                       double* __restrict__ part) {
    __shared__ double red[4][64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + lane;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_split, i1 = min(mp, i0 + rows_per_split);
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
                                                       double ls2, int64_t m, int64_t mp, double* __restrict__ A) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= mp) return;
    A[i * mp + j] = (i < m && j < m) ? G[i * m + j] + ls2 * K[i * m + j] : 0.0;
}

// C = A op(B), all n x n row-major (n a multiple of 64), f64 MFMA, 64 x 64 output tile per workgroup, K in stages of 32
// through LDS.  TB = false: C = A B (B staged k-major);  TB = true: C = A B^T (B staged row-major like A).
constexpr int GK = 32;
constexpr int GLR = GK + 2;   // row-major stage stride: [64 rows][32 k]
constexpr int GLK = JP + 16;  // k-major stage stride:   [32 k][64 cols]
template <bool TB>
__global__ __launch_bounds__(256) void gemm64_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                     double* __restrict__ C, int64_t n) {
    __shared__ double sa[64 * GLR];
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
    for (int64_t k0 = 0; k0 < n; k0 += GK) {
        const double* ap = A + (i0 + rr) * n + k0 + rc;
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = ap[q];
        if (TB) {
            const double* bp = B + (j0 + rr) * n + k0 + rc;
#pragma unroll
            for (int q = 0; q < 8; ++q) vb[q] = bp[q];
        } else {
            const double* bp = B + (k0 + kr) * n + j0 + kc;
#pragma unroll
            for (int q = 0; q < 8; ++q) vb[q] = bp[q];
        }
        __syncthreads();  // the previous stage has been consumed
#pragma unroll
        for (int q = 0; q < 8; ++q) sa[rr * GLR + rc + q] = va[q];
        if (TB) {
#pragma unroll
            for (int q = 0; q < 8; ++q) sb[rr * GLR + rc + q] = vb[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) sb[kr * GLK + kc + q] = vb[q];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            double fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                fa[a] = sa[(wr + a * 16 + li) * GLR + kk + lk];  // A[i][k]
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
                C[(i0 + row) * n + j0 + col] = acc[a][b][r];
            }
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
    static const int target = [] {
        const char* e = std::getenv("MVF_JAC_GRAM_WGS");  // developer knob: workgroups per Gram launch (default 512)
        return e ? std::max(64, atoi(e)) : 512;
    }();
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
        hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(mp, 4)), dim3(256), 0, st, Y, mp, m, R, nrhs, sig2, T);
        hipLaunchKernelGGL(jac_scale_kernel, dim3(1), dim3(256), 0, st, sig2, mp, cq.scal, rcond, T, einfo + 6);
        hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)cdiv(mp, 64), (unsigned)p.bsplit), dim3(256), 0, st, Y, mp, m,
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
    const dim3 ggrid((unsigned)(mp / 64), (unsigned)(mp / 64));
    if (warm) {
        hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mp), dim3(256), 0, st, G, K,
                           lambda_sigma2, m, mp, aux);
        hipLaunchKernelGGL(gemm64_kernel<true>, ggrid, dim3(256), 0, st, aux, basis, Y, mp);   // T1 = A Wt^T
        hipLaunchKernelGGL(gemm64_kernel<false>, ggrid, dim3(256), 0, st, basis, Y, aux, mp);  // A' = Wt T1
        MVF_LAUNCH_CHECK();
        if (int rc = chol_factor_mat(st, aux, mp, shift, m, workspace, &cp, info)) return rc;
    } else if (int rc = chol_factor(st, G, K, lambda_sigma2, shift, nullptr, m, 0, workspace, &cp, info)) {
        return rc;
    }
    int hinfo = 0;
    MVF_CHECK_HIP(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, st));
    MVF_CHECK_HIP(hipStreamSynchronize(st));
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
        MVF_CHECK_HIP(hipMemcpyAsync(&hrot, rot, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        MVF_CHECK_HIP(hipStreamSynchronize(st));
        if (hrot == 0) break;
    }

    if (warm) {
        // back to the original coordinates: row i of Y (= sigma_i x eigenvector i of A') -> Y Wt
        hipLaunchKernelGGL(gemm64_kernel<false>, ggrid, dim3(256), 0, st, Y, basis, aux, mp);
        MVF_LAUNCH_CHECK();
        MVF_CHECK_HIP(hipMemcpyAsync(Y, aux, (size_t)mp * mp * sizeof(double), hipMemcpyDeviceToDevice, st));
    }

    // 3. truncated minimum-norm solve
    hipLaunchKernelGGL(jac_rowstat_kernel, dim3((unsigned)cdiv(mp, 4)), dim3(256), 0, st, Y, mp, m, R, nrhs, sig2, T);
    hipLaunchKernelGGL(jac_scale_kernel, dim3(1), dim3(256), 0, st, sig2, mp, cp.scal, rcond, T, einfo);
    hipLaunchKernelGGL(jac_back_kernel, dim3((unsigned)cdiv(mp, 64), (unsigned)p.bsplit), dim3(256), 0, st, Y, mp, m, T,
                       p.rows_per_split, part);
    hipLaunchKernelGGL(jac_back_reduce_kernel, dim3((unsigned)cdiv(m * nrhs, 256)), dim3(256), 0, st, part, p.bsplit, m,
                       nrhs, C);
    if (basis)
        hipLaunchKernelGGL(basis_extract_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mp), dim3(256), 0, st, Y, sig2,
                           mp, basis);
    MVF_LAUNCH_CHECK();
    const double hs[1] = {(double)sweeps + (hrot != 0 ? 0.5 : 0.0)};  // x.5 = sweep cap hit before convergence
    MVF_CHECK_HIP(hipMemcpyAsync(einfo, hs, sizeof(double), hipMemcpyHostToDevice, st));
    MVF_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}
