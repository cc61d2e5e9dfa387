// con_K: materialised Gaussian RBF kernel  K[i, j] = exp(-beta ||x_i - y_j||^2)   (n x m row-major)
//
// Reference: dynamo `con_K` / in-tree twin spateo/tdr/morphometrics/morphofield/gaussian_process.py:16-36.
// Roofline: HBM-WRITE bound (algorithmic bytes = s*(n*m + n*d + m*d)); ~7 VALU + 1 v_exp per element.
//
// Mapping: a lane owns groups of VEC consecutive columns (16 bytes: 4 floats / 2 doubles) so a wave stores 1 KiB
// contiguous per row with one dwordx4 store; see the store-pattern note at the kernel.
#include "mvf_common.h"

namespace mvf {

// Store pattern (measured with tools/hbm_write_probe2.hip on MI355X): ONE workgroup per CU, each streaming a long
// CONTIGUOUS region with plain (temporal) 16-byte stores reaches 7.1 TB/s; the same bytes from 512+ workgroups or with
// non-temporal stores reach 4.2-5.5 TB/s (hipMemset: 6.3).  So the grid is `nblocks` ~ #CUs persistent workgroups and
// workgroup b writes the rows [b*rows_per_block, ...) - a contiguous slab of the row-major output - row after row.
// A lane owns PASSES groups of VEC consecutive columns (columns (p*256 + lane)*VEC ...), keeping their control points
// (pre-scaled by sqrt(beta*log2e)) in registers; the row coordinates are wave-uniform (scalar loads).
constexpr int CONK_THREADS = 1024;  // 16 waves per CU: 256 column lanes x 4 row phases (hides the per-row scalar loads)
template <typename T, int D, int VEC, int PASSES>
__global__ __launch_bounds__(CONK_THREADS) void conk_kernel(const T* __restrict__ x, int64_t n, const T* __restrict__ y,
                                                   int64_t m, int64_t col0, T s /* sqrt(beta*log2e) */,
                                                   T* __restrict__ K, int64_t rows_per_block, int dyn_d) {
    const int d = D > 0 ? D : dyn_d;
    constexpr int DMAX = D > 0 ? D : 8;
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t i1 = min(i0 + rows_per_block, n);
    T c[PASSES][VEC][DMAX];
    int64_t j0[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        j0[p] = col0 + ((int64_t)p * 256 + (threadIdx.x & 255)) * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int k = 0; k < DMAX; ++k) c[p][v][k] = (k < d && j0[p] + v < m) ? y[(j0[p] + v) * d + k] * s : T(0);
    }
    const bool vec_ok = (m % VEC) == 0;  // rows stay 16-byte aligned
    constexpr int RSTEP = CONK_THREADS / 256;
    // software pipeline: the coordinates of the NEXT row are in flight while this row is computed and stored (the row
    // index depends on the wave, so the compiler uses a vector load + vmcnt(0); un-prefetched it stalled every row)
    T xn[DMAX];
    {
        const int64_t i = i0 + (threadIdx.x >> 8);
#pragma unroll
        for (int k = 0; k < DMAX; ++k) xn[k] = (k < d && i < i1) ? x[i * d + k] : T(0);
    }
    for (int64_t i = i0 + (threadIdx.x >> 8); i < i1; i += RSTEP) {
        T xi[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) xi[k] = xn[k] * s;
        {
            const int64_t inext = i + RSTEP;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) xn[k] = (k < d && inext < i1) ? x[inext * d + k] : T(0);
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (j0[p] >= m) continue;
            T out[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                T e = T(0);
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    if (k < d) {
                        const T t = xi[k] - c[p][v][k];
                        e = fma(t, t, e);
                    }
                }
                out[v] = exp2_neg(-e);
            }
            T* dst = K + i * m + j0[p];
            if (vec_ok && j0[p] + VEC <= m) {
                if constexpr (sizeof(T) == 4 && VEC == 4) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<f4*>(dst) = f4{out[0], out[1], out[2], out[3]};
                } else if constexpr (sizeof(T) == 8 && VEC == 2) {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<d2*>(dst) = d2{out[0], out[1]};
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) dst[v] = out[v];
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (j0[p] + v < m) dst[v] = out[v];
            }
        }
    }
}

// D[n, :, m] = x_n - y_m  (n x d x m), the return_d=True companion (gaussian_process.py:25-29).
template <typename T>
__global__ __launch_bounds__(256) void conk_diff_kernel(const T* __restrict__ x, int64_t n, const T* __restrict__ y,
                                                        int64_t m, int d, T* __restrict__ Dout) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= m) return;
    for (int k = 0; k < d; ++k) Dout[(i * d + k) * m + j] = x[i * d + k] - y[j * d + k];
}

static int conk_blocks() {
    static int blocks = 0;
    if (blocks == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        (void)hipGetLastError();
        blocks = cus;  // one persistent workgroup per CU (see the store-pattern note above)
    }
    return blocks;
}

template <typename T, int D, int VEC, int PASSES>
static void launch_conk_p(const T* x, int64_t n, const T* y, int64_t m, int d, T s, T* K, hipStream_t st) {
    const int64_t cols_per_sweep = 256LL * VEC * PASSES;  // one sweep covers m <= 4096 (f32) / 2048 (f64) at PASSES = 4
    const int nb = (int)std::min<int64_t>(conk_blocks(), n);
    const int64_t rows = cdiv(n, nb);
    for (int64_t col0 = 0; col0 < m; col0 += cols_per_sweep)
        hipLaunchKernelGGL((conk_kernel<T, D, VEC, PASSES>), dim3((unsigned)cdiv(n, rows)), dim3(CONK_THREADS), 0, st, x,
                           n, y, m, col0, s, K, rows, d);
}

template <typename T, int D, int VEC>
static void launch_conk_d(const T* x, int64_t n, const T* y, int64_t m, int d, T s, T* K, hipStream_t st) {
    // PASSES is fixed at 4 (idle passes cost nothing; an m-adaptive PASSES = 2 measured 30 % SLOWER at m = 2000)
    launch_conk_p<T, D, VEC, 4>(x, n, y, m, d, s, K, st);
}

template <typename T, int VEC>
static int launch_conk(const T* x, int64_t n, const T* y, int64_t m, int d, double beta, T* K, hipStream_t st) {
    const T s = (T)std::sqrt(beta * LOG2E);
    if (d == 3)
        launch_conk_d<T, 3, VEC>(x, n, y, m, d, s, K, st);
    else if (d == 2)
        launch_conk_d<T, 2, VEC>(x, n, y, m, d, s, K, st);
    else
        launch_conk_d<T, 0, VEC>(x, n, y, m, d, s, K, st);
    MVF_LAUNCH_CHECK();
    return 0;
}

}  // namespace mvf

using namespace mvf;

extern "C" int mvf_con_k(const void* x, int64_t n, const void* y, int64_t m, int d, double beta, void* K,
                         mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n >= 0 && m >= 0 && d >= 1 && d <= 8, "mvf_con_k: bad shape n=%lld m=%lld d=%d", (long long)n,
                (long long)m, d);
    MVF_REQUIRE(beta >= 0.0 && std::isfinite(beta), "mvf_con_k: beta must be finite and >= 0");
    if (n == 0 || m == 0) return 0;
    MVF_REQUIRE(x && y && K, "mvf_con_k: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MVF_F32)
        return launch_conk<float, 4>((const float*)x, n, (const float*)y, m, d, beta, (float*)K, st);
    if (dtype == MVF_F64)
        return launch_conk<double, 2>((const double*)x, n, (const double*)y, m, d, beta, (double*)K, st);
    return set_error("mvf_con_k: bad dtype %d", (int)dtype);
}

extern "C" int mvf_con_k_d(const void* x, int64_t n, const void* y, int64_t m, int d, double beta, void* K, void* D,
                           mvf_dtype dtype, void* stream) {
    int rc = mvf_con_k(x, n, y, m, d, beta, K, dtype, stream);
    if (rc) return rc;
    if (n == 0 || m == 0) return 0;
    MVF_REQUIRE(D, "mvf_con_k_d: null D");
    MVF_REQUIRE(n <= 65535, "mvf_con_k_d: n <= 65535 rows per call (D is n x d x m)");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(m, 256), (unsigned)n);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(conk_diff_kernel<float>, grid, dim3(256), 0, st, (const float*)x, n, (const float*)y, m,
                           d, (float*)D);
    else
        hipLaunchKernelGGL(conk_diff_kernel<double>, grid, dim3(256), 0, st, (const double*)x, n, (const double*)y,
                           m, d, (double*)D);
    MVF_LAUNCH_CHECK();
    return 0;
}
