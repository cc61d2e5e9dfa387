// Developer probe 3: what the operand-preparation VALU work of the cached Gram kernel costs next to f64 MFMAs.
// Per "k-step": 16 independent v_mfma_f64_16x16x4_f64 preceded by a VALU cluster that does NOT feed them:
//   MODE 0: nothing                  MODE 1: 11 v_cvt_f64_f32 + 2 v_mul_f64 (the shipped mix)
//   MODE 2: 2 v_mul_f64 only         MODE 3: 11 converts done with 32-bit integer ops (2 shifts + 1 add each) + 2 v_mul_f64
//   MODE 4: 11 v_cvt_f64_f32 only
//   MODE 5: as MODE 1 but the converts/multiplies PRODUCE the MFMA operands of the same k-step (RAW dependency, as shipped)
//   MODE 6: as MODE 5 but converted one k-step ahead into a second register set (software-pipelined operands)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_mix_probe.hip -o tools/mfma_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double a0, double b0) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a[2], b[8], a2[2], b2[8];
    for (int i = 0; i < 2; ++i) a[i] = a0 * (1.0 + 0.37 * i) + threadIdx.x * 1.234567e-3;
    for (int i = 0; i < 8; ++i) b[i] = b0 * (1.0 + 0.11 * i) + threadIdx.x * 7.654321e-4;
    for (int i = 0; i < 2; ++i) a2[i] = a[i];
    for (int i = 0; i < 8; ++i) b2[i] = b[i];
    float f[11];
    for (int i = 0; i < 11; ++i) f[i] = 0.5f + 0.01f * i + threadIdx.x * 1e-4f;
    double d[11], m0 = 1.0000001, m1 = 0.9999999, p = 1.0 + threadIdx.x * 1e-9;
    unsigned u[11], lo[11], hi[11];
    for (int i = 0; i < 11; ++i) { d[i] = 0; u[i] = __float_as_uint(f[i]); lo[i] = hi[i] = 0; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 11; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 11; ++i)
                asm volatile("v_lshrrev_b32 %0, 3, %2\n v_add_u32 %0, 0x38000000, %0\n v_lshlrev_b32 %1, 29, %2"
                             : "=&v"(hi[i]), "=&v"(lo[i]) : "v"(u[i]));
        }
        if (MODE == 1 || MODE == 2 || MODE == 3) {
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(m0) : "v"(p));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(m1) : "v"(p));
        }
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b[i]) : "v"(f[i]));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[8 + i]));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(p));
            }
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[0]) : "v"(f[10]));
        }
        if (MODE == 6) {  // operands for the NEXT k-step go to (a2, b2); this k-step's MFMAs read (a, b)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b2[i]) : "v"(f[i]));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a2[i]) : "v"(f[8 + i]));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2[i]) : "v"(p));
            }
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[0]) : "v"(f[10]));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i >> 3]), "v"(b[i & 7]));
        __builtin_amdgcn_s_setprio(0);
        if (MODE == 6) {  // swap register sets (compile-time renaming after unrolling by 2 would avoid the moves;
                          // here: do a second identical half-iteration with the roles exchanged)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b[i]) : "v"(f[i]));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[8 + i]));
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(p));
            }
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[1]) : "v"(f[10]));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a2[i >> 3]), "v"(b2[i & 7]));
            __builtin_amdgcn_s_setprio(0);
        }
    }
    double s = m0 + m1;
    for (int i = 0; i < 11; ++i) s += d[i] + hi[i] + lo[i];
    for (int i = 0; i < 2; ++i) s += a2[i];
    for (int i = 0; i < 8; ++i) s += b2[i];
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
    const char* names[] = {"16 MFMA only", "+ 11 v_cvt_f64_f32 + 2 v_mul_f64 (shipped mix)", "+ 2 v_mul_f64",
                           "+ 11 integer-op converts (33 int ops) + 2 v_mul_f64", "+ 11 v_cvt_f64_f32",
                           "shipped mix FEEDING the same k-step's MFMAs", "shipped mix feeding the NEXT k-step (2 register sets)"};
    double ms[7];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    ms[1] = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    ms[2] = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    ms[3] = time_ms([&] { hipLaunchKernelGGL(k<3>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    ms[4] = time_ms([&] { hipLaunchKernelGGL(k<4>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    ms[5] = time_ms([&] { hipLaunchKernelGGL(k<5>, dim3(512), dim3(256), 0, 0, d, iters, 0.7391, 1.3127); }, 3);
    ms[6] = time_ms([&] { hipLaunchKernelGGL(k<6>, dim3(512), dim3(256), 0, 0, d, iters / 2, 0.7391, 1.3127); }, 3);
    for (int i = 0; i < 7; ++i) printf("%-52s: %7.2f ms  %5.1f TF\n", names[i], ms[i], fl / ms[i] / 1e9);
    return 0;
}
