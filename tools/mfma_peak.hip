// Developer probe: sustained MFMA ceilings on this GPU (pure MFMA loops, no memory traffic).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static double time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    double* d;
    hipMalloc(&d, 8 * 256 * 4096);
    const int iters = 20000;
    for (int blocks_per_cu : {1, 2}) {
        int grid = 256 * blocks_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_f64<16>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 1.0); }, 3);
        double fl = (double)grid * 4 * iters * 16 * 2048.0;
        printf("f64 16x16x4  %d WG/CU, 16 acc: %.2f ms  %.1f TF\n", blocks_per_cu, ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, (float*)d, iters, 1.f, 1.f); }, 3);
        fl = (double)grid * 4 * iters * 4 * 4096.0;
        printf("f32 32x32x2  %d WG/CU,  4 acc: %.2f ms  %.1f TF\n", blocks_per_cu, ms, fl / ms / 1e9);
    }
    // long run to see sustained (thermal / power) clocks
    double ms = time_ms([&] { hipLaunchKernelGGL(k_f64<16>, dim3(512), dim3(256), 0, 0, d, iters * 20, 1.0, 1.0); }, 2);
    printf("f64 16x16x4 long (%.0f ms): %.1f TF\n", ms, 512.0 * 4 * iters * 20 * 16 * 2048.0 / ms / 1e9);
    return 0;
}
