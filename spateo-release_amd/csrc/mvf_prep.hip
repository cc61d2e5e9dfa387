// Preprocessing on the device: lexicographically sorted unique rows of the cell coordinates.
//
// Reference: dynamo SparseVFC step 2, `tmp_X, uid = np.unique(X, axis=0, return_index=True)` (SURVEY.md Appendix A;
// the same call in-tree: `self.nx.unique(self.coordsA, return_index=True, axis=0)`, spateo/alignment/methods/
// morpho_class.py:845).  NumPy sorts the rows lexicographically and returns, per distinct row, the index of its FIRST
// occurrence; at 8 M cells that is 2-5 s of single-threaded host time.  Here: an LSD pass over the d columns (last
// column first), each a stable device radix sort of (order-preserving 64-bit image of the double, row index) - stable,
// so equal rows stay in ascending index order and the first of each run is the first occurrence - then run-start flags
// and a stream compaction.  Everything is this file's (round 4; rounds 2 - 3 called rocPRIM's device sort / select here):
//   * radix sort: 8 bits per pass, 8 passes per column.  Per pass: per-workgroup digit histograms of 4096-element chunks
//     (LDS integer atomics) -> exclusive scan of the [digit][chunk] table (three small launches) -> stable scatter: a
//     workgroup walks its chunk in sub-tiles of 256 consecutive elements, a lane's rank among the equal digits before it is
//     a wavefront match (8 ballots) plus the counts of the lower waves;
//   * compaction: per-chunk flag counts -> the same scan -> ballot-ranked scatter.
// Bit-identical to np.unique for finite input (-0.0 and +0.0 compare equal there and are given the same key here; the host
// routes non-finite input to NumPy).
#include <cstring>

#include "mvf_common.h"

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

// ---- hand-written stable LSD radix sort of (uint64 key, int64 value) pairs and flag compaction ----------------------------
constexpr int RS_BINS = 256;          // 8 bits per pass
constexpr int RS_CHUNK = 4096;        // elements per workgroup (16 sub-tiles of 256 consecutive elements)
constexpr int SCAN_SEG = 4096;        // entries per workgroup of the table scan

// hist[bin * nblocks + block] = number of keys of this chunk whose digit is `bin`
__global__ __launch_bounds__(256) void rs_hist_kernel(const unsigned long long* __restrict__ keys, int64_t n, int shift,
                                                      int64_t nblocks, unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[RS_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * RS_CHUNK;
    for (int it = 0; it < RS_CHUNK / 256; ++it) {
        const int64_t i = i0 + it * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (RS_BINS - 1)], 1u);  // integer atomics: order-independent result
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of a uint32 table of L entries, in place, in three launches: segment sums, scan of the segment sums (one
// workgroup), per-segment exclusive scan shifted by its segment's offset.  Totals stay below 2^32 (n < 2^32 keys).
__global__ __launch_bounds__(256) void scan_segsum_kernel(const unsigned int* __restrict__ t, int64_t L,
                                                          unsigned int* __restrict__ segsum) {
    __shared__ unsigned int red[256];
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_SEG;
    unsigned int s_ = 0;
    for (int k = 0; k < SCAN_SEG / 256; ++k) {
        const int64_t i = i0 + (int64_t)threadIdx.x * (SCAN_SEG / 256) + k;
        if (i < L) s_ += t[i];
    }
    red[threadIdx.x] = s_;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) segsum[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void scan_segoff_kernel(unsigned int* __restrict__ segsum, int64_t nseg,
                                                          unsigned int* __restrict__ total) {
    // one workgroup: exclusive scan of the segment sums (each thread a contiguous range, then a scan of the 256 partials)
    __shared__ unsigned int part[256];
    const int64_t per = (nseg + 255) / 256;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = min(nseg, lo + per);
    unsigned int s_ = 0;
    for (int64_t i = lo; i < hi; ++i) s_ += segsum[i];
    part[threadIdx.x] = s_;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int k = 0; k < 256; ++k) {
            const unsigned int v = part[k];
            part[k] = run;
            run += v;
        }
        if (total) *total = run;
    }
    __syncthreads();
    unsigned int run = part[threadIdx.x];
    for (int64_t i = lo; i < hi; ++i) {
        const unsigned int v = segsum[i];
        segsum[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(unsigned int* __restrict__ t, int64_t L,
                                                         const unsigned int* __restrict__ segoff) {
    __shared__ unsigned int part[256];
    constexpr int PER = SCAN_SEG / 256;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_SEG + (int64_t)threadIdx.x * PER;
    unsigned int v[PER], s_ = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        v[k] = (i0 + k < L) ? t[i0 + k] : 0u;
        s_ += v[k];
    }
    part[threadIdx.x] = s_;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = segoff[blockIdx.x];
        for (int k = 0; k < 256; ++k) {
            const unsigned int w = part[k];
            part[k] = run;
            run += w;
        }
    }
    __syncthreads();
    unsigned int run = part[threadIdx.x];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (i0 + k < L) t[i0 + k] = run;
        run += v[k];
    }
}

// lanes of this wavefront (among `valid`) that hold the same 8-bit digit as this lane
__device__ __forceinline__ unsigned long long wave_match8(unsigned int g, bool valid) {
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const bool one = (g >> bit) & 1u;
        const unsigned long long b = __ballot(one);
        mask &= one ? b : ~b;
    }
    return mask;
}

// stable scatter of one pass: element i of the chunk goes to offs[digit][block] + (number of equal digits before it in the
// chunk).  Sub-tiles of 256 consecutive elements keep the input order: lane order inside a wavefront, wave order inside
// the sub-tile, sub-tile order inside the chunk.
__global__ __launch_bounds__(256) void rs_scatter_kernel(const unsigned long long* __restrict__ kin,
                                                         const long long* __restrict__ vin,
                                                         unsigned long long* __restrict__ kout, long long* __restrict__ vout,
                                                         int64_t n, int shift, int64_t nblocks,
                                                         const unsigned int* __restrict__ offs) {
    __shared__ unsigned int base[RS_BINS];
    __shared__ unsigned int wcnt[4][RS_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    base[tid] = offs[(int64_t)tid * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * RS_CHUNK;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int it = 0; it < RS_CHUNK / 256; ++it) {
        const int64_t i = i0 + it * 256 + tid;
        const bool valid = i < n;
        const unsigned long long key = valid ? kin[i] : 0ull;
        const long long val = valid ? vin[i] : 0ll;
        const unsigned int g = (unsigned int)((key >> shift) & (RS_BINS - 1));
        const unsigned long long same = wave_match8(g, valid) & (valid ? ~0ull : 0ull);
        const int rank = __popcll(same & lt);
        if (valid && rank == 0) wcnt[wave][g] = (unsigned int)__popcll(same);
        __syncthreads();
        if (valid) {
            unsigned int pre = 0;
            for (int w = 0; w < wave; ++w) pre += wcnt[w][g];
            const unsigned int pos = base[g] + pre + (unsigned int)rank;
            kout[pos] = key;
            vout[pos] = val;
        }
        __syncthreads();
        base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
        __syncthreads();
    }
}

// compaction: cnt[block] = number of set flags of the chunk
__global__ __launch_bounds__(256) void sel_count_kernel(const unsigned char* __restrict__ flag, int64_t n,
                                                        unsigned int* __restrict__ cnt) {
    __shared__ unsigned int red[256];
    const int64_t i0 = (int64_t)blockIdx.x * RS_CHUNK;
    unsigned int s_ = 0;
    for (int it = 0; it < RS_CHUNK / 256; ++it) {
        const int64_t i = i0 + it * 256 + threadIdx.x;
        if (i < n) s_ += flag[i] ? 1u : 0u;
    }
    red[threadIdx.x] = s_;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[blockIdx.x] = red[0];
}

// out[offs[block] + (set flags before i in the chunk)] = vals[i] for every set flag (input order kept)
__global__ __launch_bounds__(256) void sel_scatter_kernel(const long long* __restrict__ vals,
                                                          const unsigned char* __restrict__ flag, int64_t n,
                                                          const unsigned int* __restrict__ offs, long long* __restrict__ out) {
    __shared__ unsigned int wc[4];
    __shared__ unsigned int run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) run = offs[blockIdx.x];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * RS_CHUNK;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int it = 0; it < RS_CHUNK / 256; ++it) {
        const int64_t i = i0 + it * 256 + tid;
        const bool f = i < n && flag[i] != 0;
        const unsigned long long b = __ballot(f);
        if (lane == 0) wc[wave] = (unsigned int)__popcll(b);
        __syncthreads();
        if (f) {
            unsigned int pre = 0;
            for (int w = 0; w < wave; ++w) pre += wc[w];
            out[run + pre + (unsigned int)__popcll(b & lt)] = vals[i];
        }
        __syncthreads();
        if (tid == 0) run += wc[0] + wc[1] + wc[2] + wc[3];
        __syncthreads();
    }
}

__global__ void count_widen_kernel(const unsigned int* __restrict__ total, long long* __restrict__ count) { *count = (long long)*total; }

struct PrepPlan {
    int64_t nblocks, L, nseg;
    size_t off_keys_a, off_keys_b, off_idx_a, off_idx_b, off_flag, off_hist, off_seg, off_cnt, off_total, total;
};

static PrepPlan prep_plan(int64_t n) {
    PrepPlan p;
    p.nblocks = cdiv(n, RS_CHUNK);
    p.L = p.nblocks * RS_BINS;
    p.nseg = cdiv(p.L, SCAN_SEG);
    size_t o = 0;
    p.off_keys_a = o, o += align_up((size_t)n * 8, 256);
    p.off_keys_b = o, o += align_up((size_t)n * 8, 256);
    p.off_idx_a = o, o += align_up((size_t)n * 8, 256);
    p.off_idx_b = o, o += align_up((size_t)n * 8, 256);
    p.off_flag = o, o += align_up((size_t)n, 256);
    p.off_hist = o, o += align_up((size_t)p.L * 4, 256);
    p.off_seg = o, o += align_up((size_t)std::max<int64_t>(p.nseg, cdiv(p.nblocks, SCAN_SEG)) * 4 + 4, 256);
    p.off_cnt = o, o += align_up((size_t)p.nblocks * 4, 256);
    p.off_total = o, o += 256;
    p.total = o + 256;
    return p;
}

// exclusive scan of t[0 .. L) in place (segsum: >= cdiv(L, SCAN_SEG) entries of scratch; total: optional, the sum)
static void exclusive_scan_u32(hipStream_t st, unsigned int* t, int64_t L, unsigned int* segsum, unsigned int* total) {
    const int64_t nseg = cdiv(L, SCAN_SEG);
    hipLaunchKernelGGL(scan_segsum_kernel, dim3((unsigned)nseg), dim3(256), 0, st, t, L, segsum);
    hipLaunchKernelGGL(scan_segoff_kernel, dim3(1), dim3(256), 0, st, segsum, nseg, total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nseg), dim3(256), 0, st, t, L, segsum);
}

// stable sort of (keys, vals) by all 64 key bits; result in (*ka, *ia) - the buffers are swapped as the passes go
static void radix_sort_pairs_u64(hipStream_t st, const PrepPlan& p, int64_t n, unsigned long long*& ka,
                                 unsigned long long*& kb, long long*& ia, long long*& ib, unsigned int* hist,
                                 unsigned int* segsum) {
    for (int shift = 0; shift < 64; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)p.nblocks), dim3(256), 0, st, ka, n, shift, p.nblocks, hist);
        exclusive_scan_u32(st, hist, p.L, segsum, nullptr);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3((unsigned)p.nblocks), dim3(256), 0, st, ka, ia, kb, ib, n, shift,
                           p.nblocks, hist);
        std::swap(ka, kb);
        std::swap(ia, ib);
    }
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
    unsigned int* hist = (unsigned int*)(ws + p.off_hist);
    unsigned int* segsum = (unsigned int*)(ws + p.off_seg);
    unsigned int* cnt = (unsigned int*)(ws + p.off_cnt);
    unsigned int* total = (unsigned int*)(ws + p.off_total);
    MVF_REQUIRE(n < (int64_t)1 << 32, "mvf_unique_rows: more than 2^32 - 1 rows");
    const dim3 grid((unsigned)cdiv(n, 256));
    hipLaunchKernelGGL(prep_iota_kernel, grid, dim3(256), 0, st, ia, n);
    for (int c = d - 1; c >= 0; --c) {  // LSD over the columns: the first column is the primary key
        hipLaunchKernelGGL(prep_keys_kernel, grid, dim3(256), 0, st, X, n, d, c, ia, ka);
        radix_sort_pairs_u64(st, p, n, ka, kb, ia, ib, hist, segsum);
    }
    hipLaunchKernelGGL(prep_flags_kernel, grid, dim3(256), 0, st, X, n, d, ia, flag);
    hipLaunchKernelGGL(sel_count_kernel, dim3((unsigned)p.nblocks), dim3(256), 0, st, flag, n, cnt);
    exclusive_scan_u32(st, cnt, p.nblocks, segsum, total);
    hipLaunchKernelGGL(sel_scatter_kernel, dim3((unsigned)p.nblocks), dim3(256), 0, st, ia, flag, n, cnt, (long long*)uid);
    hipLaunchKernelGGL(count_widen_kernel, dim3(1), dim3(1), 0, st, total, (long long*)count);
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
