// EM-loop kernels that are NOT the Gram matrix: field application V = U C (+ residuals), E-step, quadratic form.
//
// U = con_K(x, ctrl, beta) is never materialised: every kernel regenerates K(x_n, c_m) from 16-byte cell / control
// point records.  Control points (pre-scaled by sqrt(beta*log2e)) and coefficients are staged through LDS in chunks
// and read back with broadcast ds_read_b128; a lane owns CPT cells so that VALU work (6 flops + v_exp + 3 f64 FMA per
// pair and cell) outweighs the LDS broadcast traffic.  Accumulation of V is float64 in every mode: the solve is
// ill-conditioned, C can carry large cancelling coefficients, and the kernel is a ~1 % slice of an EM step anyway.
#include "mvf_common.h"

namespace mvf {

// ----------------------------------------------------------------------------------------------------------------
// apply: V = con_K(x, ctrl) @ C ; optional residual r = ||y - V||^2 and stats[0] += sum P r
// ----------------------------------------------------------------------------------------------------------------
constexpr int APPLY_CHUNK = 512;  // control points per LDS stage

template <typename T, int CPT>
__global__ __launch_bounds__(256) void apply_kernel(const T* __restrict__ x4, int64_t n, const T* __restrict__ ctrl4,
                                                    int64_t m, T s, const double* __restrict__ C /* m x 3 */,
                                                    T* __restrict__ V4, const T* __restrict__ y4,
                                                    const T* __restrict__ P, T* __restrict__ r,
                                                    double* __restrict__ block_pr) {
    using V4T = typename Vec4<T>::type;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[APPLY_CHUNK * (sizeof(V4T) + 4 * sizeof(double))];
    V4T* sc = reinterpret_cast<V4T*>(smem_raw);                                         // scaled ctrl coords
    double4* sC = reinterpret_cast<double4*>(smem_raw + APPLY_CHUNK * sizeof(V4T));     // coefficients (c0,c1,c2,0)

    const int64_t base = ((int64_t)blockIdx.x * 256) * CPT + threadIdx.x;
    T px[CPT], py[CPT], pz[CPT];
    double v0[CPT], v1[CPT], v2[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int64_t i = base + (int64_t)c * 256;
        V4T xv = (i < n) ? reinterpret_cast<const V4T*>(x4)[i] : V4T{0, 0, 0, 0};
        px[c] = xv.x * s;
        py[c] = xv.y * s;
        pz[c] = xv.z * s;
        v0[c] = v1[c] = v2[c] = 0.0;
    }

    for (int64_t m0 = 0; m0 < m; m0 += APPLY_CHUNK) {
        const int mc = (int)min((int64_t)APPLY_CHUNK, m - m0);
        __syncthreads();
        for (int j = threadIdx.x; j < APPLY_CHUNK; j += 256) {
            if (j < mc) {
                V4T cv = reinterpret_cast<const V4T*>(ctrl4)[m0 + j];
                sc[j] = V4T{cv.x * s, cv.y * s, cv.z * s, 0};
                const double* cp = C + (m0 + j) * 3;
                sC[j] = double4{cp[0], cp[1], cp[2], 0.0};
            } else {
                sc[j] = V4T{0, 0, 0, 0};
                sC[j] = double4{0.0, 0.0, 0.0, 0.0};  // zero coefficients: padded control points contribute nothing
            }
        }
        __syncthreads();
        const int mc_pad = (mc + 3) & ~3;
#pragma unroll 4
        for (int j = 0; j < mc_pad; ++j) {
            const V4T cv = sc[j];
            const double4 cc = sC[j];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const double k = (double)kernel_value(px[c], py[c], pz[c], cv.x, cv.y, cv.z);
                v0[c] = fma(k, cc.x, v0[c]);
                v1[c] = fma(k, cc.y, v1[c]);
                v2[c] = fma(k, cc.z, v2[c]);
            }
        }
    }

    double pr = 0.0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int64_t i = base + (int64_t)c * 256;
        if (i < n) {
            reinterpret_cast<V4T*>(V4)[i] = V4T{(T)v0[c], (T)v1[c], (T)v2[c], 0};
            if (y4) {
                const V4T yv = reinterpret_cast<const V4T*>(y4)[i];
                // residual against the field as stored (rounded to T), like the reference's (Y - V) on stored V
                const double d0 = (double)yv.x - (double)(T)v0[c], d1 = (double)yv.y - (double)(T)v1[c],
                             d2 = (double)yv.z - (double)(T)v2[c];
                const double rr = d0 * d0 + d1 * d1 + d2 * d2;
                r[i] = (T)rr;
                if (P) pr += (double)P[i] * (double)(T)rr;
            }
        }
    }
    if (y4 && P && block_pr) {
        __shared__ double red[4];
        const double t = block_sum<256>(pr, red);
        if (threadIdx.x == 0) block_pr[blockIdx.x] = t;  // summed in block order by sum_partials_kernel: deterministic
    }
}

// out[k] += sum over b (in order) of partials[b * stride + k]   (one workgroup, K <= 8 columns; fixed summation order)
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, int64_t nb, int stride,
                                                           int K, double* __restrict__ out, int overwrite = 0) {
    __shared__ double red[4];
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int64_t b = threadIdx.x; b < nb; b += 256) s += partials[b * stride + k];
        const double t = block_sum<256>(s, red);
        if (threadIdx.x == 0) out[k] = overwrite ? t : out[k] + t;  // (overwrite: saves the caller a memset launch)
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------------------------
// E-step (float64 arithmetic on r)
// ----------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void estep_min_kernel(const T* __restrict__ r, int64_t n, double inv2s2,
                                                        double* __restrict__ block_min_out,
                                                        double* __restrict__ block_zero_out) {
    double mn = INFINITY, zeros = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double t1 = exp(-(double)r[i] * inv2s2);
        if (t1 == 0.0)
            zeros += 1.0;
        else
            mn = fmin(mn, t1);
    }
    __shared__ double red[4];
    const double bm = block_min<256>(mn, red);
    const double bz = block_sum<256>(zeros, red);
    if (threadIdx.x == 0) {
        block_min_out[blockIdx.x] = bm;
        block_zero_out[blockIdx.x] = bz;
    }
}

__global__ __launch_bounds__(256) void estep_min_finish(const double* __restrict__ bmin, const double* __restrict__ bzero,
                                                        int nb, double* __restrict__ mins) {
    double mn = INFINITY, z = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        mn = fmin(mn, bmin[i]);
        z += bzero[i];
    }
    __shared__ double red[4];
    const double a = block_min<256>(mn, red);
    const double b = block_sum<256>(z, red);
    if (threadIdx.x == 0) {
        mins[0] = a;
        mins[1] = b;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void estep_p_kernel(const T* __restrict__ r, int64_t n, double inv2s2, double t2,
                                                      double minP, double theta, double zero_fill_host,
                                                      const double* __restrict__ fill_dev, T* __restrict__ Pout,
                                                      double* __restrict__ block_stats) {
    // the fill for underflowed t1: a device scalar (mins[0] of mvf_estep_min, possibly MIN-all-reduced across ranks;
    // +inf = no cell has a non-zero t1) when given - no host round trip between the two E-step phases - else the host's
    double zero_fill = zero_fill_host;
    if (fill_dev) {
        const double f = *fill_dev;
        zero_fill = (f < INFINITY) ? f : 0.0;
    }
    double s_pr = 0.0, s_p = 0.0, s_pf = 0.0, s_cnt = 0.0, s_zero = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double ri = (double)r[i];
        double t1 = exp(-ri * inv2s2);
        if (t1 == 0.0) {
            t1 = zero_fill;
            s_zero += 1.0;
        }
        const double p = t1 / (t1 + t2);
        s_pr += p * ri;
        s_p += p;
        const T pf = (T)fmax(p, minP);  // the floored posterior as stored (and as the Gram / sigma2 update see it)
        Pout[i] = pf;
        s_pf += (double)pf;
        s_cnt += ((double)pf > theta) ? 1.0 : 0.0;
    }
    __shared__ double red[4];
    const double a = block_sum<256>(s_pr, red);
    const double b = block_sum<256>(s_p, red);
    const double c = block_sum<256>(s_pf, red);
    const double d = block_sum<256>(s_cnt, red);
    const double e = block_sum<256>(s_zero, red);
    if (threadIdx.x == 0) {
        double* o = block_stats + (int64_t)blockIdx.x * 5;
        o[0] = a, o[1] = b, o[2] = c, o[3] = d, o[4] = e;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// trace(C^T K C)
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quadform_kernel(const double* __restrict__ K, const double* __restrict__ C,
                                                       int64_t m, int nrhs, double* __restrict__ row_out) {
    // one block per row i:  sum_d C[i,d] * sum_j K[i,j] C[j,d]  -> row_out[i] (summed in row order afterwards)
    const int64_t i = blockIdx.x;
    double acc = 0.0;
    for (int d = 0; d < nrhs; ++d) {
        double t = 0.0;
        for (int64_t j = threadIdx.x; j < m; j += 256) t += K[i * m + j] * C[j * nrhs + d];
        acc += t * C[i * nrhs + d];
    }
    __shared__ double red[4];
    const double s = block_sum<256>(acc, red);
    if (threadIdx.x == 0) row_out[i] = s;
}

// packed upper triangle (row-major: row i holds columns i..m-1) <-> full symmetric matrix
__device__ __forceinline__ int64_t tri_offset(int64_t i, int64_t m) { return i * m - i * (i - 1) / 2; }

__global__ __launch_bounds__(256) void sym_pack_kernel(const double* __restrict__ G, int64_t m, double* __restrict__ tri) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m && j >= i) tri[tri_offset(i, m) + (j - i)] = G[i * m + j];
}

__global__ __launch_bounds__(256) void sym_unpack_kernel(const double* __restrict__ tri, int64_t m, double* __restrict__ G) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    G[i * m + j] = j >= i ? tri[tri_offset(i, m) + (j - i)] : tri[tri_offset(j, m) + (i - j)];
}

}  // namespace mvf

using namespace mvf;

extern "C" int mvf_sym_pack(const double* G, int64_t m, double* tri, void* stream) {
    MVF_REQUIRE(m >= 0 && m <= 65535, "mvf_sym_pack: bad m");
    if (m == 0) return 0;
    MVF_REQUIRE(G && tri, "mvf_sym_pack: null pointer");
    hipLaunchKernelGGL(sym_pack_kernel, dim3((unsigned)cdiv(m, 256), (unsigned)m), dim3(256), 0, (hipStream_t)stream, G, m,
                       tri);
    MVF_LAUNCH_CHECK();
    return 0;
}

extern "C" int mvf_sym_unpack(const double* tri, int64_t m, double* G, void* stream) {
    MVF_REQUIRE(m >= 0 && m <= 65535, "mvf_sym_unpack: bad m");
    if (m == 0) return 0;
    MVF_REQUIRE(G && tri, "mvf_sym_unpack: null pointer");
    hipLaunchKernelGGL(sym_unpack_kernel, dim3((unsigned)cdiv(m, 256), (unsigned)m), dim3(256), 0, (hipStream_t)stream,
                       tri, m, G);
    MVF_LAUNCH_CHECK();
    return 0;
}

extern "C" int mvf_apply(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C,
                         void* V4, const void* y4, const void* P, void* r, double* stats, double* scratch,
                         mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n >= 0 && m >= 0, "mvf_apply: bad shape");
    MVF_REQUIRE(beta >= 0.0 && std::isfinite(beta), "mvf_apply: beta must be finite and >= 0");
    if (n == 0) return 0;
    MVF_REQUIRE(x4 && V4 && (m == 0 || (ctrl4 && C)), "mvf_apply: null pointer");
    MVF_REQUIRE(!y4 || r, "mvf_apply: y4 given but r is null");
    MVF_REQUIRE(!(P && y4) || (stats && scratch), "mvf_apply: P given but stats / scratch is null");
    hipStream_t st = (hipStream_t)stream;
    const double s = std::sqrt(beta * LOG2E);
    // Cells per lane: 4 (float) / 2 (double) when that still gives every compute unit several workgroups, fewer below (50 k
    // cells at 4 per lane are 49 workgroups on 256 compute units: BASELINE config 2's field update ran on a fifth of the part).
    // A cell's sum runs over the control points in the same order whatever the choice: the same bits.
    const int cpt_max = dtype == MVF_F32 ? 4 : 2;
    int cpt = cpt_max;
    while (cpt > 1 && cdiv(n, (int64_t)256 * cpt) < 1024) cpt >>= 1;
    const int64_t nblocks = cdiv(n, (int64_t)256 * cpt);
#define MVF_APPLY_LAUNCH(T, CPT_)                                                                                              \
    hipLaunchKernelGGL((apply_kernel<T, CPT_>), dim3((unsigned)nblocks), dim3(256), 0, st, (const T*)x4, n, (const T*)ctrl4, \
                       m, (T)s, C, (T*)V4, (const T*)y4, (const T*)P, (T*)r, scratch)
    if (dtype == MVF_F32) {
        if (cpt == 4) MVF_APPLY_LAUNCH(float, 4);
        else if (cpt == 2) MVF_APPLY_LAUNCH(float, 2);
        else MVF_APPLY_LAUNCH(float, 1);
    } else if (dtype == MVF_F64) {
        if (cpt == 2) MVF_APPLY_LAUNCH(double, 2);
        else MVF_APPLY_LAUNCH(double, 1);
    } else {
        return set_error("mvf_apply: bad dtype %d", (int)dtype);
    }
#undef MVF_APPLY_LAUNCH
    MVF_LAUNCH_CHECK();
    if (y4 && P) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, scratch, nblocks, 1, 1, stats);
        MVF_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" size_t mvf_reduce_scratch_doubles(int64_t n) {
    // apply: one partial per workgroup (>= 512 cells each, or fewer than 2048 workgroups); estep_p: 5 per block (<= 2048
    // blocks); quadform: one per control point
    return (size_t)std::max<int64_t>(cdiv(std::max<int64_t>(n, 1), 512) + 16, 5 * 2048 + 16);
}

namespace {
constexpr int ESTEP_MAX_BLOCKS = 2048;  // == (MVF_ESTEP_MIN_DOUBLES - 2) / 2
}  // namespace

extern "C" int mvf_estep_min(const void* r, int64_t n, double sigma2, double* mins, mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n >= 0 && sigma2 > 0.0, "mvf_estep_min: need n >= 0 and sigma2 > 0");
    MVF_REQUIRE(mins && (n == 0 || r), "mvf_estep_min: null pointer");
    static_assert(MVF_ESTEP_MIN_DOUBLES == 2 + 2 * ESTEP_MAX_BLOCKS, "scratch size mismatch with mvf.h");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)std::min<int64_t>(ESTEP_MAX_BLOCKS, std::max<int64_t>(1, cdiv(n, 256 * 4)));
    double* bmin = mins + 2;  // block partials live behind the two results (caller-provided, see mvf.h)
    double* bzero = mins + 2 + ESTEP_MAX_BLOCKS;
    const double inv2s2 = 1.0 / (2.0 * sigma2);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(estep_min_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)r, n, inv2s2, bmin, bzero);
    else if (dtype == MVF_F64)
        hipLaunchKernelGGL(estep_min_kernel<double>, dim3(nb), dim3(256), 0, st, (const double*)r, n, inv2s2, bmin,
                           bzero);
    else
        return set_error("mvf_estep_min: bad dtype %d", (int)dtype);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(estep_min_finish, dim3(1), dim3(256), 0, st, bmin, bzero, nb, mins);
    MVF_LAUNCH_CHECK();
    return 0;
}

static int estep_p_impl(const void* r, int64_t n, double sigma2, double gamma, double a, int dy, double minP, double theta,
                        double t1_zero_fill, const double* t1_zero_fill_dev, void* P_out, double* stats, double* scratch,
                        mvf_dtype dtype, void* stream, int overwrite);

extern "C" int mvf_estep_p(const void* r, int64_t n, double sigma2, double gamma, double a, int dy, double minP,
                           double theta, double t1_zero_fill, const double* t1_zero_fill_dev, void* P_out,
                           double* stats, double* scratch, mvf_dtype dtype, void* stream) {
    return estep_p_impl(r, n, sigma2, gamma, a, dy, minP, theta, t1_zero_fill, t1_zero_fill_dev, P_out, stats, scratch, dtype,
                        stream, 0);
}

// Both phases in one call (one process: nothing has to happen between them), stats OVERWRITTEN: the head of an EM iteration
// is what the device waits for after the iteration's one host read, and every separate call there is exposed host latency.
extern "C" int mvf_estep(const void* r, int64_t n, double sigma2, double gamma, double a, int dy, double minP, double theta,
                         double* mins, void* P_out, double* stats, double* scratch, mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n >= 1, "mvf_estep: need n >= 1");
    if (int rc = mvf_estep_min(r, n, sigma2, mins, dtype, stream)) return rc;
    return estep_p_impl(r, n, sigma2, gamma, a, dy, minP, theta, 0.0, mins, P_out, stats, scratch, dtype, stream, 1);
}

static int estep_p_impl(const void* r, int64_t n, double sigma2, double gamma, double a, int dy, double minP, double theta,
                        double t1_zero_fill, const double* t1_zero_fill_dev, void* P_out, double* stats, double* scratch,
                        mvf_dtype dtype, void* stream, int overwrite) {
    MVF_REQUIRE(n >= 0 && sigma2 > 0.0 && gamma > 0.0 && gamma < 1.0 && a > 0.0 && dy >= 1,
                "mvf_estep_p: bad parameters (sigma2=%g gamma=%g a=%g dy=%d)", sigma2, gamma, a, dy);
    if (n == 0) return 0;
    MVF_REQUIRE(r && P_out && stats && scratch, "mvf_estep_p: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const double inv2s2 = 1.0 / (2.0 * sigma2);
    const double t2 = std::pow(2.0 * M_PI * sigma2, dy / 2.0) * (1.0 - gamma) / (gamma * a);
    int nb = (int)std::min<int64_t>(ESTEP_MAX_BLOCKS, std::max<int64_t>(1, cdiv(n, 256 * 4)));
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(estep_p_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)r, n, inv2s2, t2, minP,
                           theta, t1_zero_fill, t1_zero_fill_dev, (float*)P_out, scratch);
    else if (dtype == MVF_F64)
        hipLaunchKernelGGL(estep_p_kernel<double>, dim3(nb), dim3(256), 0, st, (const double*)r, n, inv2s2, t2, minP,
                           theta, t1_zero_fill, t1_zero_fill_dev, (double*)P_out, scratch);
    else
        return set_error("mvf_estep_p: bad dtype %d", (int)dtype);
    MVF_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, scratch, (int64_t)nb, 5, 5, stats, overwrite);
    MVF_LAUNCH_CHECK();
    return 0;
}

extern "C" int mvf_quadform(const double* K, const double* C, int64_t m, int nrhs, double* out, double* scratch,
                            void* stream) {
    MVF_REQUIRE(m >= 0 && nrhs >= 1, "mvf_quadform: bad shape");
    MVF_REQUIRE(out && (m == 0 || scratch), "mvf_quadform: null out / scratch");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        MVF_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
        return 0;
    }
    MVF_REQUIRE(K && C, "mvf_quadform: null pointer");
    hipLaunchKernelGGL(quadform_kernel, dim3((unsigned)m), dim3(256), 0, st, K, C, m, nrhs, scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, scratch, m, 1, 1, out, 1);
    MVF_LAUNCH_CHECK();
    return 0;
}


// out[i] = a A[i] + b B[i] + c C[i]   (B, C may be NULL = 0; float64, elementwise, in-place allowed)
namespace {
__global__ __launch_bounds__(256) void lincomb3_kernel(double* __restrict__ out, double a, const double* A, double b,
                                                       const double* B, double c, const double* C, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = a * A[i];
    if (B) v = fma(b, B[i], v);
    if (C) v = fma(c, C[i], v);
    out[i] = v;
}
}  // namespace

extern "C" int mvf_lincomb3(double* out, double a, const double* A, double b, const double* B, double c, const double* C,
                            int64_t n, void* stream) {
    MVF_REQUIRE(n >= 0, "mvf_lincomb3: negative length");
    if (n == 0) return 0;
    MVF_REQUIRE(out && A, "mvf_lincomb3: null out / A");
    hipLaunchKernelGGL(lincomb3_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, a, A, b, B,
                       c, C, n);
    MVF_LAUNCH_CHECK();
    return 0;
}
