// Coefficient solve  (G + ls2 * K + jitter * mean(diag) * I) C = R   in float64.
//
// Reference: dynamo `lstsq_solver(lhs, rhs, "scipy")` as Spateo calls it (spateo/tdr/morphometrics/morphofield/
// sparsevfc.py:110,194,250).  lhs is symmetric PSD (U^T P U + lambda sigma^2 K) and
// numerically rank deficient; the reference's gelsd truncates at eps*s_max.  Here: blocked right-looking Cholesky
// (NB = 64) of the jitter-regularised matrix.  DESIGN.md ("Solve parity") explains why the FIELD, not C, is the
// parity quantity and shows the measured noise floor of the reference itself.
//
// Structure per block column k (all kernels on one stream, no host sync):
//   potrf64    : one workgroup factors the 64x64 diagonal block (register tiled, one barrier per column) - inside the
//                syrk launch of the previous block column, by the workgroup that updated that block
//   trsm_panel : one DPP quad (4 lanes x 16 interleaved columns) per row below solves  x L_kk^T = a
//   syrk_potrf : one workgroup per 64x64 trailing tile, v_mfma_f64_16x16x4_f64 on LDS-staged panels
// The right-hand sides ride along as extra ROWS of the trapezoidal matrix, so the forward substitution L Y = R falls
// out of trsm/syrk for free; the back substitution L^T C = Y is one small kernel per block column.
#include "mvf_common.h"
#include "mvf_solve.h"
#include "mvf_chol_dev.h"

namespace mvf {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// mean of the diagonal of G + ls2*K  ->  scal[0];  scal[1] = jitter * mean
__global__ __launch_bounds__(256) void diag_mean_kernel(const double* __restrict__ G, const double* __restrict__ K,
                                                        double ls2, double jitter, int64_t m,
                                                        double* __restrict__ scal) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) s += G[i * m + i] + ls2 * K[i * m + i];
    __shared__ double red[4];
    const double t = block_sum<256>(s, red);
    if (threadIdx.x == 0) {
        const double mean = t / (double)m;
        scal[0] = mean;
        scal[1] = jitter * mean;
    }
}

// W (mr x mp, row-major):  rows < mp hold the regularised matrix (identity on the padding), rows mp.. hold R^T
__global__ __launch_bounds__(256) void chol_prepare_kernel(const double* __restrict__ G, const double* __restrict__ K,
                                                           double ls2, const double* __restrict__ scal,
                                                           const double* __restrict__ R, int64_t m, int nrhs,
                                                           int64_t mp, int64_t mr, double* __restrict__ W) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= mp) return;
    double v = 0.0;
    if (i < mp) {
        if (i < m && j < m) {
            v = G[i * m + j] + ls2 * K[i * m + j];
            if (i == j) v += scal[1];
        } else if (i == j) {
            v = 1.0;
        }
    } else {
        const int d = (int)(i - mp);
        if (d < nrhs && j < m) v = R[j * nrhs + d];
    }
    W[i * mp + j] = v;
}

// ---- the three kernels of a block column (round 6: each of them was a chain of exposed latencies - 25 + 32 + 15 us per 64
// columns, profiles/r05_small_m_timeline.md - on a part whose dependent-launch boundary costs 1.5 us) ----------------------
//
// potrf64: right-looking Cholesky of one 64 x 64 block by ONE workgroup, register tiled: thread (ty, tx) of a 16 x 16 grid
// keeps a[ty + 16 p][tx + 16 q] (p >= q: the lower triangle of tiles).  A step eliminates TWO columns: the owners of columns
// j and j + 1 have published them (still UNSCALED, column j + 1 as it stood before step j) in LDS, ONE barrier, every thread
// forms column j + 1 after step j for the rows and columns it needs - the very multiply-subtract its owners would have
// applied - and applies both rank-1 updates
// a[i][c] -= (a[i][j] / a[j][j]) a[c][j] to its registers.  What makes a step short:
//   * one LDS round trip and one barrier per two columns, and the two pivots' reciprocals as independent chains;
//   * the reciprocal of a pivot is v_rcp_f64 + two Newton steps (the IEEE divide the compiler emits is ~20 dependent
//     instructions in every step's critical chain);
//   * no predicate on the update: an entry with i <= j or c <= j is dead once its column has been published (the columns
//     are kept in LDS, the result is written from there), so whatever the update does to it is never read;
//   * the NEXT two columns are updated and published first, the other tiles after them: their multiply-subtracts overlap
//     the LDS round trip of the publication instead of preceding it.
// Columns are scaled by 1 / sqrt(d_j) at the end; rdiag gets 1 / L_jj for the triangular solves.  The column loop is rolled
// inside each 16-column group (the group index must be static for the register tile; straight-line code of a
// one-workgroup kernel is paid for in instruction-cache misses).
template <int JQ>
__device__ __forceinline__ void potrf_group(double (&r)[4][4], double (*Lu)[LDU], double (*Pub)[2][NB], double* dg, int& pb,
                                            int tx, int ty, int k, int* __restrict__ info) {
    if (tx < 2) {  // the group's first two columns (the earlier groups' updates are complete in the registers)
#pragma unroll
        for (int p = JQ; p < 4; ++p) Pub[pb][tx][ty + 16 * p] = r[p][JQ];
        if (tx == 0) {
#pragma unroll
            for (int p = JQ; p < 4; ++p) Lu[16 * JQ][ty + 16 * p] = r[p][JQ];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int jx = 0; jx < 16; jx += 2) {
        const int j = 16 * JQ + jx;
        const double* u0 = Pub[pb][0];
        const double* u1 = Pub[pb][1];
        const double d0 = u0[j], e = u0[j + 1], g = u1[j + 1];
        double c0r[4], c0c[4], c1r[4], c1c[4];
#pragma unroll
        for (int p = JQ; p < 4; ++p) {  // (every LDS read of the step is in flight before a pivot is looked at)
            c0r[p] = u0[ty + 16 * p];
            c0c[p] = u0[tx + 16 * p];
            c1r[p] = u1[ty + 16 * p];
            c1c[p] = u1[tx + 16 * p];
        }
        // The two reciprocals are INDEPENDENT chains (a dependent f64 operation costs ~35 cycles on this part, a Newton
        // reciprocal five of them: tools/clock_probe.hip): 1 / d1 = d0 / (g d0 - e^2) needs no 1 / d0.  The pivot tests
        // stay off the chain: a non-positive (or NaN) pivot is reported, the arithmetic runs on with it (the factor is
        // then garbage, as info says; nothing waits for a replacement value).
        const double t1 = fma(g, d0, -(e * e));
        const double inv0 = rcp_nr2(d0);
        const double inv1 = d0 * rcp_nr2(t1);
        const double d1 = t1 * inv0;  // the pivot of column j + 1 (for the final scaling)
        const bool bad0 = !(d0 > 0.0), bad1 = !(d1 > 0.0);  // (also catch NaN)
        if (threadIdx.x == 0) {
            if (bad0 || bad1) atomicCAS(info, 0, 1 + k * NB + j + (bad0 ? 0 : 1));
            dg[j] = bad0 ? 1.0 : d0;
            dg[j + 1] = bad1 ? 1.0 : d1;
        }
        double l0[4], l1[4];
#pragma unroll
        for (int p = JQ; p < 4; ++p) {
            l0[p] = c0r[p] * inv0;
            c1r[p] = fma(-l0[p], e, c1r[p]);
            c1c[p] = fma(-(c0c[p] * inv0), e, c1c[p]);
            l1[p] = c1r[p] * inv1;
        }
        // tile column JQ first: it holds the next two columns, which their owners publish before anybody touches the rest
#pragma unroll
        for (int p = JQ; p < 4; ++p) r[p][JQ] = fma(-l1[p], c1c[JQ], fma(-l0[p], c0c[JQ], r[p][JQ]));
        if (tx == jx + 1) {  // column j + 1 as it stood when it was the pivot column: what the final scaling needs
#pragma unroll
            for (int p = JQ; p < 4; ++p) Lu[j + 1][ty + 16 * p] = c1r[p];
        }
        if (tx == jx + 2 || tx == jx + 3) {  // (never in the group's last step: the next group publishes its own first two)
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
}

// r = the block (thread (ty, tx) holds a[ty + 16 p][tx + 16 q]);  on return Lu[c][i] = unscaled column c (rows i >= c) and
// dg[c] = the pivots; the caller scales and stores with potrf64_store.  256 threads.
__device__ __forceinline__ void potrf64(double (&r)[4][4], double (*Lu)[LDU], double (*Pub)[2][NB], double* dg, int k,
                                        int* __restrict__ info) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    int pb = 0;
    potrf_group<0>(r, Lu, Pub, dg, pb, tx, ty, k, info);
    potrf_group<1>(r, Lu, Pub, dg, pb, tx, ty, k, info);
    potrf_group<2>(r, Lu, Pub, dg, pb, tx, ty, k, info);
    potrf_group<3>(r, Lu, Pub, dg, pb, tx, ty, k, info);
}

// the factor of block column k's diagonal block to W (explicitly ZERO upper triangle: the triangular solves rely on it)
__device__ __forceinline__ void potrf64_store(const double (*Lu)[LDU], const double* dg, double* __restrict__ blk, int64_t mp,
                                              double* __restrict__ rdiag_k) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = tx + 16 * q;
        const double l = sqrt(dg[c]);
        const double rl = 1.0 / l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = ty + 16 * p;
            // (16 IEEE divisions per thread would sit on the factorisation's serial path)
            blk[(int64_t)i * mp + c] = (c < i) ? div_by(Lu[c][i], l, rl) : (c == i ? l : 0.0);
            if (c == i) rdiag_k[c] = rl;
        }
    }
}

// block column 0's diagonal block (the later ones are factored by the workgroup of syrk_potrf_kernel that updated them)
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ W, int64_t mp, int k,
                                                         double* __restrict__ rdiag, int* __restrict__ info) {
    __shared__ double Lu[NB][LDU];
    __shared__ double Pub[2][2][NB];
    __shared__ double dg[NB];
    double* blk = W + ((int64_t)k * NB) * mp + (int64_t)k * NB;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double r[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[p][q] = blk[(int64_t)(ty + 16 * p) * mp + tx + 16 * q];
    potrf64(r, Lu, Pub, dg, k, info);
    potrf64_store(Lu, dg, blk, mp, rdiag + (int64_t)k * NB);
}

// rows below the diagonal block:  x L^T = a  ->  for c: x_c = a_c / L_cc ; a_j -= x_c L_jc (j > c).
// FOUR lanes (one DPP quad) per row; lane rho of the quad holds the row's columns 4 jl + rho (jl = 0 .. 15) - interleaved, so
// that at every step c all four lanes still have live columns and the uniform instruction stream carries 16 - c / 4
// multiply-subtracts instead of 16.  Per column c the owning lane scales by the precomputed 1 / L_cc, a quad_perm DPP move
// broadcasts x_c to the quad, every lane updates its columns of index >= c / 4.  The factor sits in LDS as
// Ls[c][rho][jl] = L[4 jl + rho][c] with the diagonal and the upper triangle ZERO: a lane's operands of one step are
// consecutive (ds_read_b128), entries that must not change are multiplied by zero - no predicates - and the owner's x_c
// register keeps a_c - sum (scaled once more, bit-identically, at the end).  Fully unrolled: every register index is
// static (the round-5 kernel indexed its register tile dynamically inside a rolled loop: 32 us per launch).
__global__ __launch_bounds__(256) void trsm_panel_kernel(double* __restrict__ W, int64_t mp, int64_t mr, int k,
                                                         const double* __restrict__ rdiag) {
    __shared__ __align__(16) double Ls[NB * TLC];
    __shared__ double rd[NB];
    const double* blk = W + ((int64_t)k * NB) * mp + (int64_t)k * NB;
    const int rho = threadIdx.x & 3;
    const int64_t row = (int64_t)(k + 1) * NB + (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool live = row < mr;  // whole quads are live or not (same row)
    double* rp = W + (live ? row : (int64_t)(k + 1) * NB) * mp + (int64_t)k * NB + rho;
    double x[16];
#pragma unroll
    for (int jl = 0; jl < 16; ++jl) x[jl] = rp[4 * jl];
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = blk[(int64_t)(wave * 16 + q) * mp + lane];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = wave * 16 + q;  // L[i][c = lane]
            Ls[lane * TLC + (i & 3) * TLQ + (i >> 2)] = (lane < i) ? v[q] : 0.0;
        }
        if (threadIdx.x < NB) rd[threadIdx.x] = rdiag[(int64_t)k * NB + threadIdx.x];
    }
    __syncthreads();
    const double* ls_rho = Ls + rho * TLQ;
    double2 first[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) first[h] = reinterpret_cast<const double2*>(ls_rho)[h];
    trsm_steps<0>(x, ls_rho, rd, first, rd[0]);
    if (live) {
#pragma unroll
        for (int jl = 0; jl < 16; ++jl) rp[4 * jl] = x[jl] * rd[4 * jl + rho];
    }
}

// W[i, j] -= W[i, k] W[j, k]^T  for block rows i > k (incl. the rhs block row) and block cols k < j <= min(i, nb-1); every
// load of a workgroup (both panels and its tile of W, in the accumulator layout) is issued before the first is used.  The
// workgroup of the NEXT diagonal tile (k + 1, k + 1) - tile 0 of the launch - factors it on the spot (potrf64) instead of
// storing it: a block column costs two launches, trsm_panel_kernel and this one.
constexpr int LDP = NB + 2;  // padded LDS row stride (doubles): conflict-free ds_read_b64 of MFMA operands
__global__ __launch_bounds__(256) void syrk_potrf_kernel(double* __restrict__ W, int64_t mp, int k, int nb, int nbr,
                                                         double* __restrict__ rdiag, int* __restrict__ info) {
    // decode (i, j) from the linear tile index: i in (k, nbr), j in (k, min(i, nb-1)]
    int t = blockIdx.x, i = k + 1;
    while (true) {
        const int cnt = min(i, nb - 1) - k;
        if (t < cnt) break;
        t -= cnt;
        ++i;
    }
    const int j = k + 1 + t;
    __shared__ double sa[NB * LDP];
    __shared__ double sb[NB * LDP];
    __shared__ double Pub[2][2][NB];
    __shared__ double dg[NB];
    const double* pa = W + ((int64_t)i * NB) * mp + (int64_t)k * NB;
    const double* pb = W + ((int64_t)j * NB) * mp + (int64_t)k * NB;
    double* pc = W + ((int64_t)i * NB) * mp + (int64_t)j * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
    const bool diag = i == j;
    double va[16], vb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) va[q] = pa[(int64_t)(wave * 16 + q) * mp + lane];
    if (!diag) {
#pragma unroll
        for (int q = 0; q < 16; ++q) vb[q] = pb[(int64_t)(wave * 16 + q) * mp + lane];
    }
    f64x4 cin[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cin[a][b][r] = pc[(int64_t)(wr + a * 16 + lk + 4 * r) * mp + wc + b * 16 + li];
#pragma unroll
    for (int q = 0; q < 16; ++q) sa[(wave * 16 + q) * LDP + lane] = va[q];
    if (!diag) {
#pragma unroll
        for (int q = 0; q < 16; ++q) sb[(wave * 16 + q) * LDP + lane] = vb[q];
    }
    __syncthreads();
    const double* sbb = diag ? sa : sb;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < NB; kk += 4) {
        double fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fa[a] = sa[(wr + a * 16 + li) * LDP + kk + lk];   // A[i][k]
            fb[a] = sbb[(wc + a * 16 + li) * LDP + kk + lk];  // B[k][j] = Wj[j][k]
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (blockIdx.x != 0) {  // (tile 0 is (k + 1, k + 1): the launch has tiles only while k + 1 < nb)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pc[(int64_t)(wr + a * 16 + lk + 4 * r) * mp + wc + b * 16 + li] = cin[a][b][r] - acc[a][b][r];
        return;
    }
    // the updated diagonal tile goes through LDS into potrf64's register layout and is factored here
    __syncthreads();  // every wave is done with the panels
    double(*T)[LDU] = reinterpret_cast<double(*)[LDU]>(sa);
    double(*Lu)[LDU] = reinterpret_cast<double(*)[LDU]>(sb);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[wr + a * 16 + lk + 4 * r][wc + b * 16 + li] = cin[a][b][r] - acc[a][b][r];
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double r[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[p][q] = T[ty + 16 * p][tx + 16 * q];
    potrf64(r, Lu, Pub, dg, k + 1, info);
    potrf64_store(Lu, dg, pc, mp, rdiag + (int64_t)(k + 1) * NB);
}

// back substitution L^T C = Y, one launch per block column k (descending), k + 1 workgroups:
//   every workgroup redundantly solves the 64 x 64 triangular system  L_kk^T C_k = Y_k  (Y_k is final: all later block
//   columns have already been eliminated from it), then workgroup j < k eliminates C_k from block j,
//   Y_j -= L[k-block rows, j-block cols]^T C_k, and workgroup k stores C_k.  Yw (mp x MAXR) is the working rhs.
template <int MAXR>
__global__ __launch_bounds__(256) void bsub_step_kernel(const double* __restrict__ W, int64_t mp, int k, int nrhs,
                                                        const double* __restrict__ rdiag, double* __restrict__ Yw,
                                                        double* __restrict__ Cp) {
    __shared__ double L[NB][NB + 1];
    __shared__ double t[NB][MAXR];
    __shared__ double rd[NB];
    __shared__ double part[4][NB][MAXR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* blk = W + ((int64_t)k * NB) * mp + (int64_t)k * NB;
    {
        double v[16];  // 16 independent loads in flight per lane
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = blk[(int64_t)(wave * 16 + q) * mp + lane];
#pragma unroll
        for (int q = 0; q < 16; ++q) L[wave * 16 + q][lane] = v[q];
    }
    if (tid < NB) {
        for (int d = 0; d < nrhs; ++d) t[tid][d] = Yw[((int64_t)k * NB + tid) * MAXR + d];
        rd[tid] = rdiag[(int64_t)k * NB + tid];
    }
    // issue this workgroup's elimination-panel loads early (block (k, j), rows wave*16 .. +15, column `lane`)
    const int j = blockIdx.x;
    double lcol[16];
    if (j < k) {
        const double* lp = W + ((int64_t)k * NB + wave * 16) * mp + (int64_t)j * NB + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) lcol[q] = lp[(int64_t)q * mp];
    }
    __syncthreads();
    if (wave == 0) {
        // column-oriented back substitution across the wave: lane r owns row r of the right-hand sides; once c_cc is
        // known every lane r < cc eliminates it: y_r -= L[cc][r] c_cc  ((L^T)[r][cc] = L[cc][r], contiguous over lanes)
        double y[MAXR];
#pragma unroll
        for (int d = 0; d < MAXR; ++d) y[d] = d < nrhs ? t[lane][d] : 0.0;
        for (int cc = NB - 1; cc >= 0; --cc) {
            const double inv = rd[cc];  // 1 / L[cc][cc] from potrf_diag
            const double l = L[cc][lane];
#pragma unroll
            for (int d = 0; d < MAXR; ++d) {
                if (d < nrhs) {
                    const double c = __shfl(y[d], cc, 64) * inv;
                    if (lane == cc) y[d] = c;
                    if (lane < cc) y[d] = fma(-l, c, y[d]);
                }
            }
        }
        for (int d = 0; d < nrhs; ++d) t[lane][d] = y[d];
    }
    __syncthreads();
    if (j == k) {
        if (tid < NB)
            for (int d = 0; d < nrhs; ++d) Cp[((int64_t)k * NB + tid) * nrhs + d] = t[tid][d];
        return;
    }
    double acc[MAXR];
#pragma unroll
    for (int d = 0; d < MAXR; ++d) acc[d] = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int d = 0; d < MAXR; ++d)
            if (d < nrhs) acc[d] = fma(lcol[q], t[wave * 16 + q][d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < MAXR; ++d) part[wave][lane][d] = acc[d];
    __syncthreads();
    if (tid < NB)
        for (int d = 0; d < nrhs; ++d)
            Yw[((int64_t)j * NB + tid) * MAXR + d] -= part[0][tid][d] + part[1][tid][d] + part[2][tid][d] + part[3][tid][d];
}

// working rhs for the back substitution: Yw[c][d] = W[mp + d][c]  (the forward-substituted rhs rows)
template <int MAXR>
__global__ __launch_bounds__(256) void bsub_init_kernel(const double* __restrict__ W, int64_t mp, int nrhs,
                                                        double* __restrict__ Yw) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= mp) return;
    for (int d = 0; d < MAXR; ++d) Yw[c * MAXR + d] = d < nrhs ? W[(mp + d) * mp + c] : 0.0;
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const double* __restrict__ src, int64_t count,
                                                        double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[i] = src[i];
}

// min / max of the squared Cholesky pivots L_jj^2 (j < m) from rdiag = 1 / L_jj  ->  piv[0], piv[1]
__global__ __launch_bounds__(256) void pivot_range_kernel(const double* __restrict__ rdiag, int64_t m,
                                                          double* __restrict__ piv) {
    double lo = INFINITY, hi = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        const double l = 1.0 / rdiag[i];
        lo = fmin(lo, l * l);
        hi = fmax(hi, l * l);
    }
    __shared__ double red[4];
    const double tlo = block_min<256>(lo, red);
    const double thi = -block_min<256>(-hi, red);
    if (threadIdx.x == 0) {
        piv[0] = tlo;
        piv[1] = thi;
    }
}

// ---- m <= 128 (Spateo's stock call has M = 100 control points, sparsevfc.py:248): the WHOLE solve in ONE launch ----------
// One workgroup of 1024 threads: assemble G + ls2 K (+ jitter) into a 32 x 32 thread grid of 4 x 4 register tiles
// (thread (ty, tx) keeps a[ty + 32 p][tx + 32 q]), right-looking Cholesky exactly like potrf_diag_kernel (owners publish
// the unscaled column to LDS, ONE barrier, rank-1 update in registers), the factor to LDS, then forward and back
// substitution with one thread per (row, right-hand side): the multi-launch path costs 2 x (potrf 25 + trsm 31 + bsub 19)
// + syrk 12 + 4 small launches = 0.21 ms at m = 100 - all of it dependent-launch latency.
constexpr int SM = 128;        // padded order
constexpr int SLD = SM + 1;    // LDS leading dimension of the factor (column walks hit distinct banks)
template <int JQ>
__device__ __forceinline__ void small_potrf_group(double (&r)[4][4], double (*col)[SM], double* dg, int tx, int ty,
                                                  int* bad) {
#pragma unroll 1
    for (int jx = 0; jx < 32; ++jx) {
        const int j = 32 * JQ + jx;
        if (tx == jx) {
#pragma unroll
            for (int p = JQ; p < 4; ++p) col[j & 1][ty + 32 * p] = r[p][JQ];
        }
        __syncthreads();
        const double* cb = col[j & 1];
        double d = cb[j];
        if (!(d > 0.0)) {  // also NaN: remember the first one, go on with a harmless pivot
            if (threadIdx.x == 0 && *bad == 0) *bad = 1 + j;
            d = 1.0;
        }
        if (threadIdx.x == 0) dg[j] = d;
        const double inv = 1.0 / d;
        double ci[4], cc[4];
#pragma unroll
        for (int p = JQ; p < 4; ++p) {
            ci[p] = cb[ty + 32 * p] * inv;
            cc[p] = cb[tx + 32 * p];
        }
#pragma unroll
        for (int p = JQ; p < 4; ++p)
#pragma unroll
            for (int q = JQ; q < 4; ++q) {
                const double u = r[p][q] - ci[p] * cc[q];
                r[p][q] = (ty + 32 * p > j && tx + 32 * q > j) ? u : r[p][q];
            }
    }
}

__global__ __launch_bounds__(1024) void solve_small_kernel(const double* __restrict__ G, const double* __restrict__ K,
                                                           double ls2, double jitter, const double* __restrict__ R,
                                                           int m, int nrhs, double* __restrict__ C,
                                                           int* __restrict__ info, double* __restrict__ pivots) {
    extern __shared__ __align__(16) double small_sm[];
    double* Lm = small_sm;                                            // SM x SLD
    double(*col)[SM] = reinterpret_cast<double(*)[SM]>(Lm + SM * SLD);  // 2 x SM
    double* dg = Lm + SM * SLD + 2 * SM;                              // SM
    double* yb = dg + SM;                                             // 2 x 8
    double* red = yb + 16;                                            // 16
    __shared__ int bad;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    if (tid == 0) bad = 0;
    // jitter * mean(diag)
    double jit = 0.0;
    if (jitter > 0.0) {
        double sdiag = 0.0;
        for (int i = tid; i < m; i += 1024) sdiag += G[(int64_t)i * m + i] + ls2 * K[(int64_t)i * m + i];
        const double t = block_sum<1024>(sdiag, red);
        if (tid == 0) red[0] = jitter * t / (double)m;
        __syncthreads();
        jit = red[0];
    }
    __syncthreads();
    double r[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 32 * p, j = tx + 32 * q;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < m && j < m) {
                v = G[(int64_t)i * m + j] + ls2 * K[(int64_t)i * m + j];
                if (i == j) v += jit;
            }
            r[p][q] = v;
        }
    small_potrf_group<0>(r, col, dg, tx, ty, &bad);
    small_potrf_group<1>(r, col, dg, tx, ty, &bad);
    small_potrf_group<2>(r, col, dg, tx, ty, &bad);
    small_potrf_group<3>(r, col, dg, tx, ty, &bad);
    __syncthreads();
    // L = (unscaled lower triangle) / sqrt(d_j);  the diagonal holds 1 / L_jj for the substitutions
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 32 * p, j = tx + 32 * q;
            const double rs = 1.0 / sqrt(dg[j]);
            if (j < i) Lm[i * SLD + j] = r[p][q] * rs;
            else if (j == i) Lm[i * SLD + j] = rs;
        }
    if (tid == 0) {
        info[0] = bad;
        if (pivots) {
            double lo = INFINITY, hi = 0.0;
            for (int j = 0; j < m; ++j) {
                lo = fmin(lo, dg[j]);
                hi = fmax(hi, dg[j]);
            }
            pivots[0] = lo;
            pivots[1] = hi;
        }
    }
    __syncthreads();
    // substitutions: thread (i, d) = (tid >> 3, tid & 7) carries entry (i, d) of the right-hand side
    const int i = tid >> 3, d = tid & 7;
    double v = (i < m && d < nrhs) ? R[(int64_t)i * nrhs + d] : 0.0;
    for (int k = 0; k < SM; ++k) {  // L y = R
        if (i == k) {
            v *= Lm[k * SLD + k];
            yb[(k & 1) * 8 + d] = v;
        }
        __syncthreads();
        if (i > k) v = fma(-Lm[i * SLD + k], yb[(k & 1) * 8 + d], v);
    }
    __syncthreads();
    for (int k = SM - 1; k >= 0; --k) {  // L^T c = y
        if (i == k) {
            v *= Lm[k * SLD + k];
            yb[(k & 1) * 8 + d] = v;
        }
        __syncthreads();
        if (i < k) v = fma(-Lm[k * SLD + i], yb[(k & 1) * 8 + d], v);
    }
    if (i < m && d < nrhs) C[(int64_t)i * nrhs + d] = v;
}
constexpr size_t SOLVE_SMALL_LDS = (size_t)(SM * SLD + 2 * SM + SM + 16 + 16) * sizeof(double);

static inline int64_t solve_mp(int64_t m) { return cdiv(m, NB) * NB; }

size_t chol_workspace_bytes(int64_t m, int nrhs) {
    if (m <= 0) return 0;
    const int64_t mp = solve_mp(m), mr = mp + NB;
    return align_up((size_t)mr * mp * sizeof(double), 256) + align_up((size_t)mp * std::max(nrhs, 1) * sizeof(double), 256) +
           align_up((size_t)mp * 8 * sizeof(double), 256) + align_up((size_t)mp * sizeof(double), 256) + 256;
}

// Lays out the workspace, builds W = [G + ls2 K + shift mean(diag) I ; R^T] and factors it in place (lower triangle of
// the first mp rows = L, rows mp.. = the forward-substituted right-hand sides).  Asynchronous on `st`; info[0] = 0 or
// 1 + index of the first non-positive pivot.
void chol_layout(int64_t m, int nrhs, void* workspace, CholPlan* pl) {
    const int64_t mp = solve_mp(m), mr = mp + NB;
    pl->mp = mp;
    pl->mr = mr;
    pl->nb = (int)(mp / NB);
    pl->nbr = pl->nb + 1;
    pl->W = (double*)workspace;
    pl->Cp = (double*)((char*)workspace + align_up((size_t)mr * mp * sizeof(double), 256));
    pl->Yw = (double*)((char*)pl->Cp + align_up((size_t)mp * std::max(nrhs, 1) * sizeof(double), 256));
    pl->rdiag = (double*)((char*)pl->Yw + align_up((size_t)mp * 8 * sizeof(double), 256));
    pl->scal = (double*)((char*)pl->rdiag + align_up((size_t)mp * sizeof(double), 256));
}

// the factorisation proper: W (prepared by one of the chol_prepare kernels) -> L in its lower triangle
static int chol_run(hipStream_t st, CholPlan* pl, int* info) {
    const int64_t mp = pl->mp, mr = pl->mr;
    const int nb = pl->nb, nbr = pl->nbr;
    double* W = pl->W;
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, st, W, mp, 0, pl->rdiag, info);
    for (int k = 0; k < nb; ++k) {
        const int64_t rows_below = mr - (int64_t)(k + 1) * NB;
        hipLaunchKernelGGL(trsm_panel_kernel, dim3((unsigned)cdiv(rows_below, 64)), dim3(256), 0, st, W, mp, mr, k,
                           pl->rdiag);
        // tiles: sum over i in (k, nbr) of (min(i, nb-1) - k); tile 0 = the next diagonal block, factored in the same launch
        int64_t ntiles = 0;
        for (int i = k + 1; i < nbr; ++i) ntiles += std::min(i, nb - 1) - k;
        if (ntiles > 0)
            hipLaunchKernelGGL(syrk_potrf_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, W, mp, k, nb, nbr, pl->rdiag,
                               info);
    }
    MVF_LAUNCH_CHECK();
    return 0;
}

int chol_factor(hipStream_t st, const double* G, const double* K, double ls2, double shift, const double* R, int64_t m,
                int nrhs, void* workspace, CholPlan* pl, int* info) {
    chol_layout(m, nrhs, workspace, pl);
    const int64_t mp = pl->mp, mr = pl->mr;
    MVF_REQUIRE(mr <= 65535, "coefficient solve: m too large (%lld)", (long long)m);
    MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
    hipLaunchKernelGGL(diag_mean_kernel, dim3(1), dim3(256), 0, st, G, K, ls2, shift, m, pl->scal);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(chol_prepare_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mr), dim3(256), 0, st, G, K, ls2,
                       pl->scal, R, m, nrhs, mp, mr, pl->W);
    MVF_LAUNCH_CHECK();
    return chol_run(st, pl, info);
}

// same for a matrix that is already assembled: A (leading m x m of a row-major array with leading dimension ld)
__global__ __launch_bounds__(256) void mat_diag_mean_kernel(const double* __restrict__ A, int64_t ld, double shift,
                                                            int64_t m, double* __restrict__ scal) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) s += A[i * ld + i];
    __shared__ double red[4];
    const double t = block_sum<256>(s, red);
    if (threadIdx.x == 0) {
        const double mean = t / (double)m;
        scal[0] = mean;
        scal[1] = shift * mean;
    }
}

__global__ __launch_bounds__(256) void chol_prepare_mat_kernel(const double* __restrict__ A, int64_t ld,
                                                               const double* __restrict__ scal, int64_t m, int64_t mp,
                                                               double* __restrict__ W, int ident,
                                                               const int* __restrict__ order = nullptr,
                                                               const int* __restrict__ oflag = nullptr) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= mp) return;
    double v = 0.0;
    if (i < mp) {
        if (i < m && j < m) {
            int64_t oi = i, oj = j;
            if (order != nullptr && oflag[0] != 0) {  // A[order[i]][order[j]]: the matrix in a pivot order (identity order when
                oi = order[i];                         // the caller's flag says the order is not to be trusted)
                oj = order[j];
                if (oi < 0 || oi >= m) oi = i;
                if (oj < 0 || oj >= m) oj = j;
            }
            v = A[oi * ld + oj];
            if (i == j && scal != nullptr) v += scal[1];
        } else if (i == j) {
            v = 1.0;
        }
    } else if (ident && i - mp == j) {
        v = 1.0;  // the identity as right-hand-side rows: they leave the factorisation as L^-T
    }
    W[i * mp + j] = v;
}

// The whole factorisation of a matrix of order <= 64 in ONE launch (the Gram matrices of the deflation block's Cholesky-QR
// and its 64 x 64 Rayleigh-Ritz matrix: memset + prepare + potrf + trsm were four dependent launches, 31 us): load with
// identity padding, potrf64, and - inverse != 0 - the 64 identity rows through the quad substitution of trsm_panel_kernel,
// one quad per row: W (128 x 64) = [L ; L^-T] exactly as the blocked path leaves it.
__global__ __launch_bounds__(256) void chol64_kernel(const double* __restrict__ A, int64_t ld, int m, double* __restrict__ W,
                                                     double* __restrict__ rdiag, int* __restrict__ info, int inverse) {
    __shared__ double Lu[NB][LDU];
    __shared__ double Pub[2][2][NB];
    __shared__ double dg[NB];
    __shared__ __align__(16) double Ls[NB * TLC];
    __shared__ double rd[NB];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double r[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 16 * p, c = tx + 16 * q;
            r[p][q] = (i < m && c < m) ? A[(int64_t)i * ld + c] : (i == c ? 1.0 : 0.0);
        }
    potrf64(r, Lu, Pub, dg, 0, info);
    potrf64_store(Lu, dg, W, NB, rdiag);
    if (!inverse) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = tx + 16 * q;
        const double l = sqrt(dg[c]);
        const double rl = 1.0 / l;
        if (ty == 0) rd[c] = rl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = ty + 16 * p;
            Ls[c * TLC + (i & 3) * TLQ + (i >> 2)] = (c < i) ? div_by(Lu[c][i], l, rl) : 0.0;
        }
    }
    __syncthreads();
    const int rho = threadIdx.x & 3, row = threadIdx.x >> 2;
    double x[16];
#pragma unroll
    for (int jl = 0; jl < 16; ++jl) x[jl] = (4 * jl + rho == row) ? 1.0 : 0.0;
    const double* ls_rho = Ls + rho * TLQ;
    double2 first[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) first[h] = reinterpret_cast<const double2*>(ls_rho)[h];
    trsm_steps<0>(x, ls_rho, rd, first, rd[0]);
    double* rp = W + (int64_t)(NB + row) * NB + rho;
#pragma unroll
    for (int jl = 0; jl < 16; ++jl) rp[4 * jl] = x[jl] * rd[4 * jl + rho];
}

int chol_factor_mat(hipStream_t st, const double* A, int64_t ld, double shift, int64_t m, void* workspace, CholPlan* pl,
                    int* info) {
    chol_layout(m, 0, workspace, pl);
    const int64_t mp = pl->mp, mr = pl->mr;
    MVF_REQUIRE(mr <= 65535, "coefficient solve: m too large (%lld)", (long long)m);
    MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
    hipLaunchKernelGGL(mat_diag_mean_kernel, dim3(1), dim3(256), 0, st, A, ld, shift, m, pl->scal);
    hipLaunchKernelGGL(chol_prepare_mat_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mr), dim3(256), 0, st, A, ld,
                       pl->scal, m, mp, pl->W, 0);
    MVF_LAUNCH_CHECK();
    return chol_run(st, pl, info);
}

// Cholesky AND the inverse of the factor in one pass: the right-hand sides ride along as extra rows (x L^T = a), so mp rows
// holding the identity come out as L^-T (row j = column j of L^-1): pl->W + mp * mp, leading dimension mp, upper triangular.
// No shift; info is NOT reset (the caller chains several factorisations and reads it once).
size_t chol_inv_workspace_bytes(int64_t m) {
    if (m <= 0) return 0;
    const int64_t mp = solve_mp(m);
    return align_up((size_t)2 * mp * mp * sizeof(double), 256) + align_up((size_t)mp * sizeof(double), 256) + 256;
}

// inverse = 0: the factor only (one block row of zero right-hand sides keeps the kernels' trapezoidal layout)
// order / oflag (may be NULL): factor A[order][order] instead (oflag[0] == 0: the order is not to be trusted - identity)
int chol_factor_mat_inv(hipStream_t st, const double* A, int64_t ld, int64_t m, void* workspace, CholPlan* pl, int* info,
                        int inverse, const int* order, const int* oflag) {
    const int64_t mp = solve_mp(m), mr = inverse ? 2 * mp : mp + NB;
    MVF_REQUIRE(mr <= 65535, "coefficient solve: m too large (%lld)", (long long)m);
    pl->mp = mp;
    pl->mr = mr;
    pl->nb = (int)(mp / NB);
    pl->nbr = (int)(mr / NB);
    pl->W = (double*)workspace;
    pl->Cp = pl->Yw = nullptr;
    pl->rdiag = (double*)((char*)workspace + align_up((size_t)mr * mp * sizeof(double), 256));
    pl->scal = (double*)((char*)pl->rdiag + align_up((size_t)mp * sizeof(double), 256));
    if (mp == NB && order == nullptr) {  // one block: one launch
        hipLaunchKernelGGL(chol64_kernel, dim3(1), dim3(256), 0, st, A, ld, (int)m, pl->W, pl->rdiag, info, inverse);
        MVF_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(chol_prepare_mat_kernel, dim3((unsigned)cdiv(mp, 256), (unsigned)mr), dim3(256), 0, st, A, ld,
                       (const double*)nullptr, m, mp, pl->W, inverse, order, oflag);
    MVF_LAUNCH_CHECK();
    return chol_run(st, pl, info);
}

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_solve_workspace_bytes(int64_t m, int nrhs) { return chol_workspace_bytes(m, nrhs); }

extern "C" int mvf_solve(const double* G, const double* K, double lambda_sigma2, double jitter, const double* R,
                         int64_t m, int nrhs, double* C, int* info, double* pivots, void* workspace,
                         size_t workspace_bytes, void* stream) {
    MVF_REQUIRE(m >= 0 && nrhs >= 1 && nrhs <= 8, "mvf_solve: need m >= 0 and 1 <= nrhs <= 8 (got m=%lld nrhs=%d)",
                (long long)m, nrhs);
    MVF_REQUIRE(info, "mvf_solve: null info");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        MVF_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
        return 0;
    }
    MVF_REQUIRE(G && K && R && C, "mvf_solve: null pointer");
    MVF_REQUIRE(std::isfinite(lambda_sigma2) && lambda_sigma2 >= 0.0 && jitter >= 0.0, "mvf_solve: bad regularisation");
    const size_t need = mvf_solve_workspace_bytes(m, nrhs);
    MVF_REQUIRE(workspace && workspace_bytes >= need, "mvf_solve: workspace too small (%zu < %zu)", workspace_bytes, need);
    const bool small_off = debug_opt(DBG_SOLVE_SMALL_OFF) != 0;  // developer option: the blocked multi-launch path at every m
    if (m <= SM && !small_off) {
        auto kern = solve_small_kernel;
        // (per call: the attribute belongs to the current device's copy of the kernel; the call costs ~1 us)
        // A device that refuses 135 KB of dynamic LDS (a part with 64 KB of LDS, a restricted carve-out) takes the blocked
        // path below, which handles every m.
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)SOLVE_SMALL_LDS) == hipSuccess) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(1024), SOLVE_SMALL_LDS, st, G, K, lambda_sigma2, jitter, R, (int)m, nrhs,
                               C, info, pivots);
            MVF_LAUNCH_CHECK();
            return 0;
        }
        (void)hipGetLastError();
    }
    CholPlan pl;
    if (int rc = chol_factor(st, G, K, lambda_sigma2, jitter, R, m, nrhs, workspace, &pl, info)) return rc;
    if (pivots) {
        hipLaunchKernelGGL(pivot_range_kernel, dim3(1), dim3(256), 0, st, pl.rdiag, m, pivots);
        MVF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bsub_init_kernel<8>, dim3((unsigned)cdiv(pl.mp, 256)), dim3(256), 0, st, pl.W, pl.mp, nrhs, pl.Yw);
    for (int k = pl.nb - 1; k >= 0; --k)
        hipLaunchKernelGGL(bsub_step_kernel<8>, dim3((unsigned)(k + 1)), dim3(256), 0, st, pl.W, pl.mp, k, nrhs,
                           pl.rdiag, pl.Yw, pl.Cp);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)cdiv(m * nrhs, 256)), dim3(256), 0, st, pl.Cp, m * nrhs, C);
    MVF_LAUNCH_CHECK();
    return 0;
}
