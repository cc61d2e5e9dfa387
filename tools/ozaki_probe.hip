// Developer probe (GPU): compute-side ceiling of an exact-integer ("Ozaki") Gram accumulation on gfx950.
//
// Idea under test (VERDICT r2 item 7 / DESIGN 4.1(c)): write each operand entry  a = sqrt(P) K  as NS signed 7-bit slices
// a = sum_k 2^(-7 (k + 1)) a_k  (a_k int8, a structured rounding of the operand), form the NS^2 slice products with
// v_mfma_i32_32x32x32_i8 (exact int32 sums), keep one int32 accumulator per weight group w = k + l (2 NS - 1 groups: they
// carry different powers of two and cannot share an accumulator), and fold the groups into float64 accumulators every F
// k-blocks (F x 32 cells; int32 stays exact for F x 32 x 4 x 127^2 < 2^31, i.e. F <= 1040).
//
// This probe keeps all operands in registers (no memory traffic at all) and measures
//   (1) the plain int8 MFMA issue rate,
//   (2) the rate of the full inner loop: NS^2 MFMAs per 32-cell block + the int32 -> float64 fold of (2 NS - 1) x 16
//       values per lane every F blocks,
// expressed in "float64-equivalent TFLOP/s": one 32-cell block of a 32 x 32 output tile = 2 x 32^3 flop of full-precision
// work, whatever the number of slice products behind it.  Compare with 77.8 TF measured for v_mfma_f64_16x16x4_f64.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ozaki_probe.hip -o tools/ozaki_probe && tools/ozaki_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void mfma_only(const v4i* __restrict__ in, v16i* __restrict__ out, int iters) {
    v4i a = in[threadIdx.x], b = in[threadIdx.x + 256];
    v16i acc[4];
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 16; ++e) acc[q][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[q], 0, 0, 0);
    }
    v16i s = acc[0] + acc[1] + acc[2] + acc[3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NS, int F>
__global__ __launch_bounds__(256) void ozaki_loop(const v4i* __restrict__ in, double* __restrict__ out, int blocks) {
    constexpr int NG = 2 * NS - 1;
    v4i a[NS], b[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        a[k] = in[threadIdx.x + 64 * k];
        b[k] = in[threadIdx.x + 64 * (k + NS)];
    }
    v16i g[NG];
    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
#pragma unroll
    for (int w = 0; w < NG; ++w)
#pragma unroll
        for (int e = 0; e < 16; ++e) g[w][e] = 0;
    for (int blk = 0; blk < blocks; blk += F) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int l = 0; l < NS; ++l) g[k + l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[l], g[k + l], 0, 0, 0);
            // keep the operands "new" for the compiler without touching memory
#pragma unroll
            for (int k = 0; k < NS; ++k) a[k][0] ^= blk + f;
        }
#pragma unroll
        for (int w = 0; w < NG; ++w) {
            const double scale = __builtin_ldexp(1.0, -7 * (w + 2));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[e] = __builtin_fma((double)g[w][e], scale, acc[e]);
                g[w][e] = 0;
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename KernelFn>
static float time_ms(KernelFn launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int wgs = cus * 8;  // 8 workgroups of 4 waves per CU queued: every SIMD always has work
    std::vector<int> h(4 * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int)(i * 2654435761u);
    v4i* in;
    double* out;
    CHECK(hipMalloc(&in, h.size() * sizeof(int)));
    CHECK(hipMalloc(&out, (size_t)wgs * 256 * sizeof(v16i)));
    CHECK(hipMemcpy(in, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    const double waves = (double)wgs * 4;
    {
        const int iters = 4096;
        const float ms = time_ms([&] { hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(256), 0, 0, in, (v16i*)out, iters); }, 5);
        const double ops = waves * iters * 4.0 * 2.0 * 32 * 32 * 32;
        printf("{\"probe\": \"int8 mfma 32x32x32 only\", \"ms\": %.3f, \"TOPS\": %.1f}\n", ms, ops / ms / 1e9);
    }
    const int blocks = 4096;
#define RUN(NS, F)                                                                                                         \
    {                                                                                                                       \
        const float ms = time_ms(                                                                                           \
            [&] { hipLaunchKernelGGL((ozaki_loop<NS, F>), dim3(wgs), dim3(256), 0, 0, in, out, blocks); }, 5);              \
        const double flops = waves * blocks * 2.0 * 32 * 32 * 32;                                                           \
        printf("{\"probe\": \"ozaki loop\", \"slices\": %d, \"fold_every_blocks\": %d, \"ms\": %.3f, \"f64_equiv_TFLOPs\": " \
               "%.1f, \"int8_TOPS\": %.1f}\n",                                                                              \
               NS, F, ms, flops / ms / 1e9, flops * NS * NS / ms / 1e9);                                                    \
    }
    RUN(4, 1) RUN(4, 2) RUN(4, 4) RUN(4, 8) RUN(4, 16) RUN(3, 4) RUN(3, 16)
    return 0;
}
