// Developer probe: which streaming-store pattern reaches the hipMemset write rate (6.1 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: grid-stride, 1 float4 per iteration.  MODE 1: each block owns a contiguous chunk, UNR float4 per lane per
// iteration (lane-interleaved).  MODE 2: like 1 with non-temporal stores.
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void fill(f4* out, size_t n4) {
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    if (MODE == 0) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = v;
    } else {
        const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
        const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
        for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * UNR) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const size_t j = i + (size_t)u * 256;
                if (j < b1) {
                    if (MODE == 2) __builtin_nontemporal_store(v, out + j); else out[j] = v;
                }
            }
        }
    }
}
int main() {
    const size_t n4 = (size_t)2000000 * 2000 / 4;
    f4* K; hipMalloc(&K, n4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-40s %7.3f ms  %7.1f GB/s\n", name, ms, n4 * 16.0 / ms / 1e6);
    };
    for (int g : {256, 512, 1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "chunked temporal unr4 grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((fill<1, 4>), dim3(g), dim3(256), 0, 0, K, n4); });
        snprintf(nm, 64, "chunked temporal unr8 grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((fill<1, 8>), dim3(g), dim3(256), 0, 0, K, n4); });
        snprintf(nm, 64, "chunked nontemp  unr4 grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL((fill<2, 4>), dim3(g), dim3(256), 0, 0, K, n4); });
    }
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(K, 0, n4 * 16, 0); });
    return 0;
}
