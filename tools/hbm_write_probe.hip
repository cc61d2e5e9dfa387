// Developer probe: HBM write ceiling (pure 16-byte streaming stores, temporal vs non-temporal) vs the con_K kernel.
#include "../spateo-release_amd/csrc/mvf_lib.hip"
#include "../spateo-release_amd/csrc/mvf_conk.hip"
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(f4* out, size_t n4) {
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
int main() {
    const int64_t n = 2000000, m = 2000;
    float *x, *c, *K;
    hipMalloc(&x, n * 12); hipMalloc(&c, m * 12); hipMalloc(&K, (size_t)n * m * 4);
    hipMemset(x, 0, n * 12); hipMemset(c, 0, m * 12);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)n * m * 4;
    auto timeit = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    for (int g : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "fill temporal  grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_kernel<false>, dim3(g), dim3(256), 0, 0, (f4*)K, (size_t)n * m / 4); });
        snprintf(nm, 64, "fill nontemp   grid %d", g);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_kernel<true>, dim3(g), dim3(256), 0, 0, (f4*)K, (size_t)n * m / 4); });
    }
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(K, 0, (size_t)n * m * 4, 0); });
    timeit("mvf_con_k f32 2M x 2000 d=3", [&] { mvf_con_k(x, n, c, m, 3, 1e-5, K, MVF_F32, nullptr); });
    return 0;
}
