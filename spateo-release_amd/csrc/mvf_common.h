// Shared host/device helpers for libmvf (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include "../../include/mvf.h"

namespace mvf {

// ---- error channel (thread-local message + int status across the C ABI) ----
char* err_buf();
int set_error(const char* fmt, ...);
// Developer options (process-wide, set only through mvf_debug_option; the library never reads the environment): which
// kernel variant / plan a launch takes in A/B measurements and in the tests that compare the variants.  0 = default.
enum DebugOpt { DBG_CONK_FORM = 0, DBG_SLICE_LEN, DBG_SOLVE_SMALL_OFF, DBG_LR_TIMING, DBG_LR_NO_DEFLATE, DBG_DEFL_BLOCK, DBG_DEFL_APPS, DBG_LR_NO_DIRECT, DBG_DIRECT_ACCEPT, DBG_COUNT };
long long debug_opt(DebugOpt which);
// compute units of the CURRENT HIP device (looked up once per device index; 256 if the query fails)
int device_cu_count();

#define MVF_CHECK_HIP(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return ::mvf::set_error("%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define MVF_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) return ::mvf::set_error(__VA_ARGS__); \
    } while (0)

#define MVF_LAUNCH_CHECK() MVF_CHECK_HIP(hipGetLastError())

constexpr int WAVE = 64;

// log2(e): exp(-beta d^2) == exp2(-(beta*log2e) d^2); coordinates are pre-scaled by sqrt(beta*log2e)
constexpr double LOG2E = 1.4426950408889634074;

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
    using type = float4;
};
template <>
struct Vec4<double> {
    using type = double4;
};

// exp2 of a non-positive argument
__device__ __forceinline__ float exp2_neg(float e) { return __builtin_amdgcn_exp2f(e); }  // v_exp_f32
__device__ __forceinline__ double exp2_neg(double e) { return exp2(e); }

// THE Gaussian kernel value, on coordinates pre-scaled by sqrt(beta*log2e).  Every kernel (Gram, rhs, apply, eval,
// con_K) must produce bit-identical K(x, c) for the same inputs: the coefficients C are fitted to the U implied by the
// Gram/rhs kernels and carry large cancelling entries, so a 1e-6 relative disagreement between "the U that was fitted"
// and "the U that is applied" shows up as a 1e-2 error in V = U C.  Hence one function, explicit fma, and the whole
// library is built with -ffp-contract=off so the compiler cannot fuse the scaling multiply into the subtraction in
// one kernel and not in another.
template <typename T>
__device__ __forceinline__ T kernel_value(T px, T py, T pz, T cx, T cy, T cz) {
    const T dx = px - cx, dy = py - cy, dz = pz - cz;
    const T e = fma(dz, dz, fma(dy, dy, dx * dx));
    return exp2_neg(-e);
}

// ---- wave / block reductions (wave64 shuffles, then LDS across waves) ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}

// Sum `v` over a block of NT threads (NT multiple of 64, <= 1024); result valid in thread 0.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sm /* >= NT/64 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) t += sm[i];
    }
    return t;
}
template <int NT>
__device__ __forceinline__ double block_min(double v, double* sm) {
    v = wave_min(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double t = INFINITY;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) t = fmin(t, sm[i]);
    }
    return t;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace mvf
