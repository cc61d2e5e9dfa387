// Developer probe: effective shader clock and dependent-chain latencies inside a ONE-workgroup kernel (the regime of the
// coefficient solve's serial kernels), alone and beside a kernel that keeps the other CUs busy.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ab/clock_probe tools/clock_probe.hip && tools/ab/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void chain_kernel(double* out, unsigned long long* t, int n, int mode) {
    double x = out[threadIdx.x], a = 1.0000001, b = 1e-9;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    if (mode == 0) {
        for (int i = 0; i < n; ++i) x = fma(x, a, b);                   // dependent f64 FMA chain
    } else if (mode == 1) {
        for (int i = 0; i < n; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;  // dependent v_rcp_f64 + add
    } else if (mode == 2) {
        __shared__ double s[256];
        s[threadIdx.x] = x;
        for (int i = 0; i < n; ++i) {                                    // LDS round trip + barrier per step
            __syncthreads();
            x = s[(threadIdx.x + 1) & 255] * a;
            __syncthreads();
            s[threadIdx.x] = x;
        }
    } else if (mode == 3) {
        for (int i = 0; i < n; ++i) x = sqrt(x + 2.0);                    // IEEE sqrt chain
    } else if (mode == 4) {
        float f = (float)x + 1.0f;
        for (int i = 0; i < n; ++i) f = fmaf(f, 1.0000001f, 1e-9f);       // dependent f32 FMA chain
        x = f;
    } else if (mode == 5) {
        float f = (float)x + 1.0f;
        for (int i = 0; i < n; ++i) f = __builtin_amdgcn_rsqf(f) + 1.5f;   // dependent v_rsq_f32 + add
        x = f;
    } else if (mode == 6) {
#pragma unroll 8
        for (int i = 0; i < n; ++i) x = fma(x, a, b);                     // f64 FMA chain, unrolled x 8
    } else if (mode == 7) {
        double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
        for (int i = 0; i < n; ++i) {                                      // four independent f64 FMA chains
            y0 = fma(y0, a, b);
            y1 = fma(y1, a, b);
            y2 = fma(y2, a, b);
            y3 = fma(y3, a, b);
        }
        x = y0 + y1 + y2 + y3;
    } else if (mode == 9) {
        double y[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) y[q] = x + q;
#pragma unroll 4
        for (int i = 0; i < n; ++i) {                                      // 16 independent f64 FMA chains: issue rate
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = fma(y[q], a, b);
        }
        x = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) x += y[q];
    } else if (mode == 10) {
        float y[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) y[q] = (float)x + q;
#pragma unroll 4
        for (int i = 0; i < n; ++i) {                                      // 16 independent f32 FMA chains: issue rate
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = fmaf(y[q], 1.0000001f, 1e-9f);
        }
        float z = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) z += y[q];
        x = z;
    } else {
        for (int i = 0; i < n; ++i) {                                      // f64 -> f32 -> f64 conversion round trip
            float f = (float)x;
            x = (double)f + 1.0;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

__global__ void busy_kernel(float* p, int iters) {
    float x = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0001f, 0.5f);
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    double* out;
    unsigned long long* t;
    float* busy;
    hipMalloc(&out, 256 * 8);
    hipMalloc(&t, 16);
    hipMalloc(&busy, 1024 * 256 * 4);
    hipMemset(out, 0, 256 * 8);
    hipMemset(busy, 0, 1024 * 256 * 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    const char* names[11] = {"f64 fma chain", "v_rcp_f64 + add chain", "LDS round trip + 2 barriers", "IEEE sqrt chain", "f32 fma chain", "v_rsq_f32 + add chain", "f64 fma chain unrolled", "4 independent f64 fma chains", "cvt f64->f32->f64 + add", "16 independent f64 fma (per step = 16 FMAs)", "16 independent f32 fma (per step = 16 FMAs)"};
    const int n = 20000;
    for (int loaded = 0; loaded < 2; ++loaded)
        for (int mode = 0; mode < 11; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                if (loaded) hipLaunchKernelGGL(busy_kernel, dim3(1020), dim3(256), 0, s2, busy, 4000000);
                hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(256), 0, s1, out, t, n, mode);
                hipStreamSynchronize(s1);
                unsigned long long h[2];
                hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
                hipDeviceSynchronize();
                if (rep == 2)
                    printf("%-10s %-30s: %7.1f shader cycles / step, %7.1f ns / step, effective clock %6.0f MHz\n",
                           loaded ? "beside load" : "alone", names[mode], (double)h[0] / n, (double)h[1] * 10.0 / n,
                           (double)h[0] / ((double)h[1] * 10.0) * 1000.0);
            }
        }
    return 0;
}
