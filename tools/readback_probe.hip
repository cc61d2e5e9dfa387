// Developer probe: what a small device -> host status read costs on this part, per pattern (the EM iteration and the factor form
// of the coefficient solve wait for the host at such reads).  Each pattern: a ~10 us kernel that writes the status words, then
// the read, 2000 times; reported: microseconds per (kernel + read) minus the kernel alone.
//   hipcc --offload-arch=gfx950 -O2 tools/readback_probe.hip -o tools/readback_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(double* a, int* b, int it, volatile double* hostmapped) {
    double v = a[0];
    for (int i = 0; i < 3000; ++i) v = v * 1.0000001 + 1e-9;
    if (threadIdx.x == 0) {
        a[1] = v;
        b[0] = it;
        if (hostmapped) { hostmapped[1] = v; __threadfence_system(); hostmapped[0] = (double)it; }
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double* da; int* db;
    CK(hipMalloc(&da, 64 * sizeof(double)));
    CK(hipMalloc(&db, 64 * sizeof(int)));
    CK(hipMemset(da, 0, 64 * sizeof(double)));
    char* pin;
    CK(hipHostMalloc((void**)&pin, 4096, hipHostMallocPortable));
    volatile double* mapped;
    CK(hipHostMalloc((void**)&mapped, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    double* mapped_dev;
    CK(hipHostGetDevicePointer((void**)&mapped_dev, (void*)mapped, 0));
    const int N = 2000;
    auto run = [&](const char* name, auto&& read, double* mp) -> double {
        for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, da, db, w, mp); read(w); }
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, da, db, 100 + i, mp); read(100 + i); }
        auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        printf("%-64s %8.2f us per (kernel + read)\n", name, us);
        return us;
    };
    double hs[8]; int hinfo;
    run("kernel + hipStreamSynchronize only (no read)", [&](int) { (void)hipStreamSynchronize(st); }, nullptr);
    run("2 x hipMemcpyAsync to PAGEABLE + hipStreamSynchronize (shipped)", [&](int) {
        (void)hipMemcpyAsync(hs, da, 64, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(&hinfo, db, 4, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st); }, nullptr);
    run("1 x hipMemcpyAsync to PAGEABLE + hipStreamSynchronize", [&](int) {
        (void)hipMemcpyAsync(hs, da, 64, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st); }, nullptr);
    run("2 x hipMemcpyAsync to PINNED + hipStreamSynchronize", [&](int) {
        (void)hipMemcpyAsync(pin, da, 64, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(pin + 256, db, 4, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st); memcpy(hs, pin, 64); memcpy(&hinfo, pin + 256, 4); }, nullptr);
    run("1 x hipMemcpyAsync to PINNED + hipStreamSynchronize", [&](int) {
        (void)hipMemcpyAsync(pin, da, 64, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st); memcpy(hs, pin, 64); }, nullptr);
    run("1 x hipMemcpy (synchronous) to PAGEABLE", [&](int) { (void)hipMemcpy(hs, da, 64, hipMemcpyDeviceToHost); }, nullptr);
    run("1 x hipMemcpyDtoH via hipMemcpyWithStream to PAGEABLE", [&](int) { (void)hipMemcpyWithStream(hs, da, 64, hipMemcpyDeviceToHost, st); }, nullptr);
    run("kernel writes COHERENT mapped host memory + hipStreamSynchronize", [&](int) { (void)hipStreamSynchronize(st); hs[0] = mapped[1]; }, mapped_dev);
    run("kernel writes COHERENT mapped host memory, host SPINS on the flag", [&](int it) {
        while (mapped[0] != (double)it) { } hs[0] = mapped[1]; }, mapped_dev);
    return 0;
}
