// con_K: materialised Gaussian RBF kernel  K[i, j] = exp(-beta ||x_i - y_j||^2)   (n x m row-major)
//
// Reference: dynamo `con_K` / in-tree twin spateo/tdr/morphometrics/morphofield/gaussian_process.py:16-36.
// Roofline: HBM-WRITE bound (algorithmic bytes = s*(n*m + n*d + m*d)); ~7 VALU + 1 v_exp per element.
//
// Mapping: a lane owns VEC consecutive columns (16 bytes: 4 floats / 2 doubles) so a wave stores 1 KiB contiguous
// per row with one dwordx4 store; a block of 256 lanes covers 256*VEC columns and loops over ROWS rows, keeping its
// VEC control points (pre-scaled by sqrt(beta*log2e)) in registers.  Row coordinates are wave-uniform -> scalar
// loads.  Output is streamed with non-temporal stores (never re-read by this kernel).
#include "mvf_common.h"

#include <string>

namespace mvf {

template <typename T, int D, int VEC>
__global__ __launch_bounds__(256) void conk_kernel(const T* __restrict__ x, int64_t n, const T* __restrict__ y,
                                                   int64_t m, T s /* sqrt(beta*log2e) */, T* __restrict__ K,
                                                   int rows_per_block, int dyn_d) {
    const int d = D > 0 ? D : dyn_d;
    const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t i1 = min(i0 + rows_per_block, n);
    constexpr int DMAX = D > 0 ? D : 8;
    T c[VEC][DMAX];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int k = 0; k < DMAX; ++k) c[v][k] = (k < d && j0 + v < m) ? y[(j0 + v) * d + k] * s : T(0);
    if (j0 >= m) return;
    const bool full = (j0 + VEC <= m) && ((m % VEC) == 0);  // aligned vector store possible
    for (int64_t i = i0; i < i1; ++i) {
        T xi[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) xi[k] = (k < d) ? x[i * d + k] * s : T(0);  // wave-uniform
        T out[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            T e = T(0);
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                if (k < d) {
                    const T t = xi[k] - c[v][k];
                    e = fma(t, t, e);
                }
            }
            out[v] = exp2_neg(-e);
        }
        T* dst = K + i * m + j0;
        if (full) {
            if constexpr (sizeof(T) == 4 && VEC == 4) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 o = {out[0], out[1], out[2], out[3]};
                __builtin_nontemporal_store(o, reinterpret_cast<f4*>(dst));
            } else if constexpr (sizeof(T) == 8 && VEC == 2) {
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 o = {out[0], out[1]};
                __builtin_nontemporal_store(o, reinterpret_cast<d2*>(dst));
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) dst[v] = out[v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (j0 + v < m) dst[v] = out[v];
        }
    }
}

// Flat streaming form (the default for d = 3, m % VEC == 0, control points that fit LDS).  K is one contiguous array of
// n m elements: a workgroup owns contiguous, aligned CHUNKS of it (256 lanes x VEC elements x U passes = 16 KB) and walks
// them with a grid stride, so at any moment the resident workgroups write ONE contiguous window of memory that moves
// linearly through the buffer - the access pattern of a device memset (6.2 - 6.3 TB/s on this part), where the 2-D form
// above writes 4 KB pieces strided by the row length (5.2 - 5.6 TB/s; rows of 8000 B also leave every other row's wave
// segments straddling 128-byte lines).  Price: a lane's vector no longer has a wave-uniform row, so (row, column) come
// from one division per chunk (wave-uniform) plus a small per-lane quotient, the row's coordinates are a per-lane cached
// load and the control points are read from an LDS copy (structure of arrays, pre-scaled; one 16-byte LDS read per
// coordinate and vector).  Same arithmetic per element as the 2-D form (and as kernel_value): bit-identical output.
template <typename T, int VEC, int U>
__global__ __launch_bounds__(256) void conk_flat_kernel(const T* __restrict__ x, int64_t n, const T* __restrict__ y,
                                                        int m, int mp /* m rounded up to VEC */, T s,
                                                        T* __restrict__ K, int64_t nchunks, float rcp_m) {
    extern __shared__ __align__(16) unsigned char conk_smem[];
    T* cx = reinterpret_cast<T*>(conk_smem);
    T* cy = cx + mp;
    T* cz = cy + mp;
    for (int j = threadIdx.x; j < m; j += 256) {
        cx[j] = y[3 * j] * s;
        cy[j] = y[3 * j + 1] * s;
        cz[j] = y[3 * j + 2] * s;
    }
    __syncthreads();
    constexpr int CH = 256 * VEC * U;  // elements per chunk
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t e0 = c * CH;           // wave-uniform
        const int64_t i0 = e0 / m;
        const unsigned j0 = (unsigned)(e0 - i0 * m);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = (unsigned)(u * 256 + threadIdx.x) * VEC;
            const unsigned jj = j0 + off;    // < m + CH < 2^24: exact in float
            unsigned q = (unsigned)((float)jj * rcp_m);
            int r = (int)(jj - q * (unsigned)m);
            if (r < 0) {
                r += m;
                --q;
            } else if (r >= m) {
                r -= m;
                ++q;
            }
            const int64_t i = i0 + q;
            if (i >= n) continue;
            const T px = x[3 * i] * s, py = x[3 * i + 1] * s, pz = x[3 * i + 2] * s;
            const vec_t vx = *reinterpret_cast<const vec_t*>(cx + r);
            const vec_t vy = *reinterpret_cast<const vec_t*>(cy + r);
            const vec_t vz = *reinterpret_cast<const vec_t*>(cz + r);
            vec_t o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T t0 = px - vx[v], t1 = py - vy[v], t2 = pz - vz[v];
                const T e = fma(t2, t2, fma(t1, t1, fma(t0, t0, T(0))));
                o[v] = exp2_neg(-e);
            }
            __builtin_nontemporal_store(o, reinterpret_cast<vec_t*>(K + e0 + off));
        }
    }
}

// Row-contiguous form: a workgroup owns ROWS consecutive FULL rows of K - one contiguous piece of memory, written front
// to back - and covers a SPAN of RS consecutive rows (RS m elements) in NP passes of 256 lanes x VEC elements, so that a
// lane keeps the control points of its NP x VEC span slots in registers (no LDS, no per-lane index arithmetic in the
// loop) and the row coordinates stay wave-uniform scalar loads as in the 2-D form.  RS is chosen by the host so that a
// span is a multiple of 64 bytes (the HBM burst): every wave's 1 KB segment then stays burst-aligned although a single
// row (3000 float32 = 12 000 bytes) is not - measured 4.5 TB/s with RS = 1 on such rows against 5.65 on aligned ones.
// Same arithmetic per element: bit-identical output.
template <typename T, int VEC, int NP, int RS>
__global__ __launch_bounds__(256) void conk_rows_kernel(const T* __restrict__ x, int64_t n, const T* __restrict__ y,
                                                        int m, T s, T* __restrict__ K, int spans_per_block) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int span = RS * m;
    T cx[NP][VEC], cy[NP][VEC], cz[NP][VEC];
    int roff[NP], col[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = (p * 256 + (int)threadIdx.x) * VEC;  // slot in the span
        roff[p] = c < span ? c / m : RS;                  // RS = "outside the span"
        col[p] = c - roff[p] * m;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const bool ok = c < span;
            const int j = ok ? col[p] + v : 0;
            cx[p][v] = ok ? y[3 * j] * s : T(0);
            cy[p][v] = ok ? y[3 * j + 1] * s : T(0);
            cz[p][v] = ok ? y[3 * j + 2] * s : T(0);
        }
    }
    const int64_t nspans = (n + RS - 1) / RS;
    const int64_t s0 = (int64_t)blockIdx.x * spans_per_block;
    const int64_t s1 = min(s0 + spans_per_block, nspans);
    for (int64_t sp = s0; sp < s1; ++sp) {
        const int64_t i0 = sp * RS;
        T px[RS], py[RS], pz[RS];  // wave-uniform
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int64_t i = min(i0 + r, n - 1);
            px[r] = x[3 * i] * s;
            py[r] = x[3 * i + 1] * s;
            pz[r] = x[3 * i + 2] * s;
        }
        T* base = K + i0 * (int64_t)m;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (roff[p] >= RS || i0 + roff[p] >= n) continue;
            T qx = px[0], qy = py[0], qz = pz[0];
#pragma unroll
            for (int r = 1; r < RS; ++r) {
                const bool hit = roff[p] == r;
                qx = hit ? px[r] : qx;
                qy = hit ? py[r] : qy;
                qz = hit ? pz[r] : qz;
            }
            vec_t o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T t0 = qx - cx[p][v], t1 = qy - cy[p][v], t2 = qz - cz[p][v];
                const T e = fma(t2, t2, fma(t1, t1, fma(t0, t0, T(0))));
                o[v] = exp2_neg(-e);
            }
            __builtin_nontemporal_store(o, reinterpret_cast<vec_t*>(base + (p * 256 + (int)threadIdx.x) * VEC));
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

// LDS the flat form may use per workgroup (two workgroups per CU stay resident at the limit)
constexpr size_t CONK_FLAT_LDS_MAX = 80 * 1024;

template <typename T, int VEC>
static int launch_conk(const T* x, int64_t n, const T* y, int64_t m, int d, double beta, T* K, hipStream_t st) {
    const T s = (T)std::sqrt(beta * LOG2E);
    // developer option conk_form: 1 = the row-contiguous form, 2 = the flat-chunk form, 3 = the 2-D row-block form
    static const char* const forms[4] = {nullptr, "rows", "flat", "2d"};
    const long long fk = debug_opt(DBG_CONK_FORM);
    const char* knob = (fk >= 1 && fk <= 3) ? forms[fk] : nullptr;
    // Default: the row-contiguous form whenever a burst-aligned span of RS <= 4 rows fits its register budget (8 passes),
    // else the flat form (float32) / the 2-D form (float64: its flat form is VALU-bound); profiles/r03_conk_ab*.json.
    int rs = 1;
    while (rs < 4 && ((int64_t)rs * m * (int64_t)sizeof(T)) % 64 != 0) rs *= 2;
    const bool rows_fit = d == 3 && m % VEC == 0 && m >= 64 && ((int64_t)rs * m * (int64_t)sizeof(T)) % 64 == 0 &&
                          (int64_t)rs * m <= 256 * VEC * 8;
    const std::string form = knob ? std::string(knob) : std::string(rows_fit ? "rows" : (sizeof(T) == 4 ? "flat" : "2d"));
    const bool legacy = form != "flat";
    if (form == "rows" && rows_fit) {
        // ~128 KB of contiguous output per workgroup, at most 16 rows (measured 8 / 16 / 32 / 64 rows per workgroup: within
        // 3 % of each other, profiles/r03_conk_ab_rowspans.json)
        const int64_t row_bytes = m * (int64_t)sizeof(T);
        const int rows_auto = (int)std::min<int64_t>(16, std::max<int64_t>(rs, (131072 / row_bytes) / rs * rs));
        const int rows_pb = rows_auto;
        const int spans_pb = std::max(1, rows_pb / rs);
        const dim3 grid((unsigned)cdiv(cdiv(n, rs), spans_pb));
        // (32 bytes per lane and pass - a 2 x VEC vector store - was measured too: 1.9 - 2.3 TB/s, the compiler does not emit
        // two adjacent 16-byte stores for it; profiles/r03_conk_ab_wide.json)
#define MVF_CONK_ROWS_CASE(VECV, NPV, RSV)                                                                                 \
    hipLaunchKernelGGL((conk_rows_kernel<T, VECV, NPV, RSV>), grid, dim3(256), 0, st, x, n, y, (int)m, s, K, spans_pb)
#define MVF_CONK_ROWS_NP(VECV, RSV)                                  \
    switch ((int)cdiv((int64_t)rs * m, 256 * VECV)) {                \
        case 1: MVF_CONK_ROWS_CASE(VECV, 1, RSV); break;             \
        case 2: MVF_CONK_ROWS_CASE(VECV, 2, RSV); break;             \
        case 3: MVF_CONK_ROWS_CASE(VECV, 3, RSV); break;             \
        case 4: MVF_CONK_ROWS_CASE(VECV, 4, RSV); break;             \
        case 5: MVF_CONK_ROWS_CASE(VECV, 5, RSV); break;             \
        case 6: MVF_CONK_ROWS_CASE(VECV, 6, RSV); break;             \
        case 7: MVF_CONK_ROWS_CASE(VECV, 7, RSV); break;             \
        default: MVF_CONK_ROWS_CASE(VECV, 8, RSV); break;            \
    }
#define MVF_CONK_ROWS_RS(VECV)          \
    if (rs == 1) {                      \
        MVF_CONK_ROWS_NP(VECV, 1)       \
    } else if (rs == 2) {               \
        MVF_CONK_ROWS_NP(VECV, 2)       \
    } else {                            \
        MVF_CONK_ROWS_NP(VECV, 4)       \
    }
        MVF_CONK_ROWS_RS(VEC)
#undef MVF_CONK_ROWS_RS
#undef MVF_CONK_ROWS_NP
#undef MVF_CONK_ROWS_CASE
        MVF_LAUNCH_CHECK();
        return 0;
    }
    constexpr int U = 4;
    const size_t lds = (size_t)3 * m * sizeof(T);
    if (!legacy && d == 3 && m % VEC == 0 && lds <= CONK_FLAT_LDS_MAX && m >= 64) {
        constexpr int CH = 256 * VEC * U;
        const int64_t nchunks = cdiv(n * m, CH);
        const int cus = device_cu_count();  // cached per device: hipGetDeviceProperties is a slow host call
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
        const int64_t grid = std::min<int64_t>(nchunks, (int64_t)cus * per_cu);
        auto kern = conk_flat_kernel<T, VEC, U>;
        if (lds > 64 * 1024)
            MVF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONK_FLAT_LDS_MAX));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, x, n, y, (int)m, (int)m, s, K, nchunks,
                           1.0f / (float)m);
        MVF_LAUNCH_CHECK();
        return 0;
    }
    const int rows = 32;
    dim3 grid((unsigned)cdiv(m, 256 * VEC), (unsigned)cdiv(n, rows));
    if (d == 3)
        hipLaunchKernelGGL((conk_kernel<T, 3, VEC>), grid, dim3(256), 0, st, x, n, y, m, s, K, rows, d);
    else if (d == 2)
        hipLaunchKernelGGL((conk_kernel<T, 2, VEC>), grid, dim3(256), 0, st, x, n, y, m, s, K, rows, d);
    else
        hipLaunchKernelGGL((conk_kernel<T, 0, VEC>), grid, dim3(256), 0, st, x, n, y, m, s, K, rows, d);
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
    MVF_REQUIRE(cdiv(n, 32) <= 65535 * 1024LL, "mvf_con_k: n too large");
    hipStream_t st = (hipStream_t)stream;
    // grid.y is limited to 65535: chunk rows
    const int64_t max_rows = 65535LL * 32;
    for (int64_t r0 = 0; r0 < n; r0 += max_rows) {
        const int64_t nr = (n - r0 < max_rows) ? n - r0 : max_rows;
        int rc;
        if (dtype == MVF_F32)
            rc = launch_conk<float, 4>((const float*)x + r0 * d, nr, (const float*)y, m, d, beta,
                                       (float*)K + r0 * m, st);
        else if (dtype == MVF_F64)
            rc = launch_conk<double, 2>((const double*)x + r0 * d, nr, (const double*)y, m, d, beta,
                                        (double*)K + r0 * m, st);
        else
            return set_error("mvf_con_k: bad dtype %d", (int)dtype);
        if (rc) return rc;
    }
    return 0;
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
