// Preprocessing on the device: lexicographically sorted unique rows of the cell coordinates.
//
// Reference: dynamo SparseVFC step 2, `tmp_X, uid = np.unique(X, axis=0, return_index=True)` (SURVEY.md Appendix A;
// the same call in-tree: `self.nx.unique(self.coordsA, return_index=True, axis=0)`, spateo/alignment/methods/
// morpho_class.py:845).  NumPy sorts the rows lexicographically and returns, per distinct row, the index of its FIRST
// occurrence; at 8 M cells that is 2-5 s of single-threaded host time.  Here: an LSD pass over the d columns (last
// column first), each a stable device radix sort of (order-preserving 64-bit image of the double, row index) - stable,
// so equal rows stay in ascending index order and the first of each run is the first occurrence - then run-start flags
// and a stream compaction.  The radix sort and the compaction are rocPRIM's device primitives (header-only part of
// ROCm; this is a one-off O(N) preprocessing step, not a kernel of the EM loop); the key transform, gather, flag and
// row-gather kernels are this file's.  Bit-identical to np.unique for finite input (-0.0 and +0.0 compare equal there
// and are given the same key here; the host routes non-finite input to NumPy).
#include <cstring>

#include "mvf_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace mvf {

__device__ __forceinline__ unsigned long long ordered_key(double x) {
    if (x == 0.0) x = 0.0;  // -0.0 -> +0.0: they are equal for np.unique
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}

__global__ __launch_bounds__(256) void prep_iota_kernel(long long* __restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = i;
}

__global__ __launch_bounds__(256) void prep_keys_kernel(const double* __restrict__ X, int64_t n, int d, int c,
                                                        const long long* __restrict__ idx,
                                                        unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = ordered_key(X[idx[i] * d + c]);
}

// flag[i] = 1 when sorted row i starts a run (differs from sorted row i - 1 in some column; value comparison, so that
// -0.0 == 0.0 like NumPy)
__global__ __launch_bounds__(256) void prep_flags_kernel(const double* __restrict__ X, int64_t n, int d,
                                                         const long long* __restrict__ idx,
                                                         unsigned char* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned char f = (i == 0);
    if (i > 0) {
        const double* a = X + idx[i] * d;
        const double* b = X + idx[i - 1] * d;
        for (int c = 0; c < d; ++c) f |= (a[c] != b[c]);
    }
    flag[i] = f;
}

__global__ __launch_bounds__(256) void prep_gather_rows_kernel(const double* __restrict__ X, int d,
                                                               const long long* __restrict__ uid,
                                                               const long long* __restrict__ count,
                                                               double* __restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= *count) return;
    for (int c = 0; c < d; ++c) rows[i * d + c] = X[uid[i] * d + c];
}

struct PrepPlan {
    size_t off_keys_a, off_keys_b, off_idx_a, off_idx_b, off_flag, off_tmp, tmp_bytes, total;
};

static PrepPlan prep_plan(int64_t n) {
    PrepPlan p;
    size_t sort_bytes = 0, sel_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                    (const long long*)nullptr, (long long*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    (void)rocprim::select(nullptr, sel_bytes, (const long long*)nullptr, (const unsigned char*)nullptr, (long long*)nullptr,
                          (long long*)nullptr, (size_t)n, (hipStream_t)0);
    p.tmp_bytes = std::max(sort_bytes, sel_bytes);
    size_t o = 0;
    p.off_keys_a = o, o += align_up((size_t)n * 8, 256);
    p.off_keys_b = o, o += align_up((size_t)n * 8, 256);
    p.off_idx_a = o, o += align_up((size_t)n * 8, 256);
    p.off_idx_b = o, o += align_up((size_t)n * 8, 256);
    p.off_flag = o, o += align_up((size_t)n, 256);
    p.off_tmp = o, o += align_up(p.tmp_bytes, 256);
    p.total = o + 256;
    return p;
}


// ---- kNN bandwidth of the control points ----------------------------------------------------------------------------
// dynamo `bandwidth_selector`: exact kNN with k = max(2, int(0.2 m)) neighbours (self included), d = mean of the k - 1
// non-self distances over all points, h = sqrt(2) d / 1.5.  One workgroup per point: all m
// squared distances into LDS (float64; the squares are accumulated coordinate by coordinate, no contraction), an
// ascending bitonic sort of the padded power-of-two array, then the sum of the square roots of ranks 1 .. k-1 (rank 0 is
// the point itself).  rowsum[i] is deterministic (fixed-order block sum); the host takes the mean.
__global__ __launch_bounds__(256) void knn_rowsum_kernel(const double* __restrict__ X, int m, int d, int k, int np2,
                                                         double* __restrict__ rowsum) {
    extern __shared__ double sd[];
    __shared__ double red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    double xi[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xi[c] = c < d ? X[(int64_t)i * d + c] : 0.0;
    for (int j = tid; j < np2; j += 256) {
        double s2 = INFINITY;
        if (j < m) {
            s2 = 0.0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < d) {
                    const double df = xi[c] - X[(int64_t)j * d + c];
                    s2 = s2 + df * df;
                }
        }
        sd[j] = s2;
    }
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < np2 / 2; t += 256) {
                const int i0 = (t / stride) * 2 * stride + (t % stride), i1 = i0 + stride;
                const bool up = (i0 & size) == 0;
                const double a = sd[i0], b = sd[i1];
                if ((a > b) == up) {
                    sd[i0] = b;
                    sd[i1] = a;
                }
            }
        }
    __syncthreads();
    double acc = 0.0;
    for (int j = 1 + tid; j < k; j += 256) acc += sqrt(sd[j]);
    const double t = block_sum<256>(acc, red);
    if (tid == 0) rowsum[i] = t;
}

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_unique_rows_workspace_bytes(int64_t n, int d) {
    if (n <= 0 || d < 1) return 0;
    return prep_plan(n).total;
}

extern "C" int mvf_unique_rows(const double* X, int64_t n, int d, int64_t* uid, double* rows, int64_t* count,
                               void* workspace, size_t workspace_bytes, void* stream) {
    MVF_REQUIRE(n >= 0 && d >= 1 && d <= 16, "mvf_unique_rows: bad shape (n=%lld d=%d)", (long long)n, d);
    MVF_REQUIRE(count, "mvf_unique_rows: null count");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        MVF_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int64_t), st));
        return 0;
    }
    MVF_REQUIRE(X && uid && rows, "mvf_unique_rows: null pointer");
    const PrepPlan p = prep_plan(n);
    MVF_REQUIRE(workspace && workspace_bytes >= p.total, "mvf_unique_rows: workspace too small (%zu < %zu)",
                workspace_bytes, p.total);
    char* ws = (char*)workspace;
    unsigned long long* ka = (unsigned long long*)(ws + p.off_keys_a);
    unsigned long long* kb = (unsigned long long*)(ws + p.off_keys_b);
    long long* ia = (long long*)(ws + p.off_idx_a);
    long long* ib = (long long*)(ws + p.off_idx_b);
    unsigned char* flag = (unsigned char*)(ws + p.off_flag);
    void* tmp = ws + p.off_tmp;
    const dim3 grid((unsigned)cdiv(n, 256));
    hipLaunchKernelGGL(prep_iota_kernel, grid, dim3(256), 0, st, ia, n);
    for (int c = d - 1; c >= 0; --c) {  // LSD over the columns: the first column is the primary key
        hipLaunchKernelGGL(prep_keys_kernel, grid, dim3(256), 0, st, X, n, d, c, ia, ka);
        size_t bytes = p.tmp_bytes;
        MVF_CHECK_HIP(rocprim::radix_sort_pairs(tmp, bytes, ka, kb, ia, ib, (size_t)n, 0, 64, st));
        std::swap(ia, ib);
    }
    hipLaunchKernelGGL(prep_flags_kernel, grid, dim3(256), 0, st, X, n, d, ia, flag);
    size_t bytes = p.tmp_bytes;
    MVF_CHECK_HIP(rocprim::select(tmp, bytes, ia, flag, (long long*)uid, (long long*)count, (size_t)n, st));
    hipLaunchKernelGGL(prep_gather_rows_kernel, grid, dim3(256), 0, st, X, d, (const long long*)uid,
                       (const long long*)count, rows);
    MVF_LAUNCH_CHECK();
    return 0;
}

extern "C" int mvf_knn_rowsum(const double* X, int64_t m, int d, int k, double* rowsum, void* stream) {
    MVF_REQUIRE(m >= 1 && m <= 8192 && d >= 1 && d <= 8, "mvf_knn_rowsum: need 1 <= m <= 8192 points of 1 <= d <= 8 (got %lld x %d)",
                (long long)m, d);
    MVF_REQUIRE(k >= 2 && k <= m, "mvf_knn_rowsum: need 2 <= k <= m (got k=%d, m=%lld)", k, (long long)m);
    MVF_REQUIRE(X && rowsum, "mvf_knn_rowsum: null pointer");
    int np2 = 2;
    while (np2 < m) np2 <<= 1;
    hipLaunchKernelGGL(knn_rowsum_kernel, dim3((unsigned)m), dim3(256), (size_t)np2 * sizeof(double), (hipStream_t)stream, X,
                       (int)m, d, k, np2, rowsum);
    MVF_LAUNCH_CHECK();
    return 0;
}


// ---- convex-hull mask -------------------------------------------------------------------------------------------------
// inside[i] = 1 iff  max over facets f of (n_f . p_i + d_f) <= tol :  a point is in a convex polytope iff it is on the inner
// side of every facet half-space (SciPy ConvexHull.equations rows are (n_f, d_f) with n_f . x + d_f <= 0 inside).
// One lane per point, the facets staged through LDS in chunks of 256; float64 throughout.
namespace {
constexpr int HULL_CHUNK = 256;
__global__ __launch_bounds__(256) void hull_mask_kernel(const double* __restrict__ pts, int64_t n,
                                                        const double* __restrict__ eq, int64_t nf, double tol,
                                                        unsigned char* __restrict__ inside) {
    __shared__ double se[HULL_CHUNK][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const double px = live ? pts[3 * i] : 0.0, py = live ? pts[3 * i + 1] : 0.0, pz = live ? pts[3 * i + 2] : 0.0;
    double worst = -INFINITY;
    for (int64_t f0 = 0; f0 < nf; f0 += HULL_CHUNK) {
        const int fc = (int)min((int64_t)HULL_CHUNK, nf - f0);
        __syncthreads();
        if ((int)threadIdx.x < fc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) se[threadIdx.x][q] = eq[(f0 + threadIdx.x) * 4 + q];
        }
        __syncthreads();
        for (int f = 0; f < fc; ++f)
            worst = fmax(worst, fma(se[f][0], px, fma(se[f][1], py, fma(se[f][2], pz, se[f][3]))));
    }
    // fmax() drops a NaN operand, so a non-finite point would leave worst = -inf and count as inside; find_simplex says -1
    if (live) inside[i] = (worst <= tol && isfinite(px) && isfinite(py) && isfinite(pz)) ? 1 : 0;
}
}  // namespace

extern "C" int mvf_hull_mask(const double* points, int64_t n, const double* equations, int64_t nfacets, double tol,
                             unsigned char* inside, void* stream) {
    MVF_REQUIRE(n >= 0 && nfacets >= 1, "mvf_hull_mask: need n >= 0 and at least one facet");
    MVF_REQUIRE(std::isfinite(tol), "mvf_hull_mask: bad tolerance");
    if (n == 0) return 0;
    MVF_REQUIRE(points && equations && inside, "mvf_hull_mask: null pointer");
    hipLaunchKernelGGL(hull_mask_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, points, n,
                       equations, nfacets, tol, inside);
    MVF_LAUNCH_CHECK();
    return 0;
}
