// Developer probe: does the 8000-byte row pitch (64-B aligned rows, wave stores straddling 128-B lines) cost bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// rows of `m4` float4; block b owns rows [b*rows, ...); lane -> float4 within the row, PASSES passes of 256 lanes
__global__ __launch_bounds__(256) void fill_rows(f4* out, long n, int m4, long rows) {
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    const long i0 = blockIdx.x * rows, i1 = i0 + rows < n ? i0 + rows : n;
    for (long i = i0; i < i1; ++i)
        for (int j = threadIdx.x; j < m4; j += 256) out[i * m4 + j] = v;
}
// flat: block b owns the same contiguous slab but lanes sweep it in aligned 4 KiB strips regardless of rows
__global__ __launch_bounds__(256) void fill_flat(f4* out, long n, int m4, long rows) {
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    const long b0 = blockIdx.x * rows * m4, b1 = (blockIdx.x * rows + rows < n ? blockIdx.x * rows + rows : n) * m4;
    for (long e = b0 + threadIdx.x; e < b1; e += 256) out[e] = v;
}
int main() {
    f4* K; hipMalloc(&K, (size_t)2000000 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m : {2000, 2048, 3000, 3072}) {
        const long n = 2000000; const int m4 = m / 4; const long rows = (n + 255) / 256;
        auto timeit = [&](const char* name, auto f) {
            f(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("m=%d %-12s %7.3f ms  %7.1f GB/s\n", m, name, ms, (double)n * m * 4 / ms / 1e6);
        };
        if ((size_t)n * m > (size_t)2000000 * 2048) { const long n2 = 1300000; (void)n2; }
        const long nn = m > 2048 ? 1300000 : n; const long rr = (nn + 255) / 256;
        auto t2 = [&](const char* name, auto f) {
            f(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("m=%d %-12s %7.3f ms  %7.1f GB/s\n", m, name, ms, (double)nn * m * 4 / ms / 1e6);
        };
        t2("rows", [&] { hipLaunchKernelGGL(fill_rows, dim3(256), dim3(256), 0, 0, K, nn, m4, rr); });
        t2("flat", [&] { hipLaunchKernelGGL(fill_flat, dim3(256), dim3(256), 0, 0, K, nn, m4, rr); });
        (void)timeit; (void)rows;
    }
    return 0;
}
