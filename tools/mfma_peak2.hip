// Developer probe 2: f64 MFMA ceiling under different operand / interleave patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = a0 * (1.0 + 0.37 * i) + threadIdx.x * 1.234567e-3;
        b[i] = b0 * (1.0 + 0.11 * i) + threadIdx.x * 7.654321e-4;
    }
    float f = threadIdx.x * 0.001f, g = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[0]), "v"(b[0]));
            if (MODE >= 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i >> 2]), "v"(b[i & 3]));
            if (MODE == 2) {  // 8 independent VALU ops between MFMAs
                asm volatile("v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %1, %1\n"
                             "v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %1, %1" : "+v"(f) : "v"(g));
            }
        }
    }
    double s = f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static double time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    double* d;
    (void)hipMalloc(&d, 8 * 256 * 4096);
    const int iters = 20000;
    const double fl = 512.0 * 4 * iters * 16 * 2048.0;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, d, iters, 1.0, 1.0); }, 3);
    printf("mode0 same A/B regs, a=b=1.0     : %.1f TF\n", fl / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    printf("mode0 same A/B regs, random-ish  : %.1f TF\n", fl / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    printf("mode1 4x4 distinct A/B           : %.1f TF\n", fl / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    printf("mode2 + 8 VALU between MFMAs     : %.1f TF\n", fl / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, d, iters, 0.0, 0.0); }, 3);
    printf("mode1 zeros-ish operands         : %.1f TF\n", fl / ms / 1e9);
    return 0;
}
