// Developer probe: times the Gram tile kernels (recompute + cached-U) on synthetic data, with ablation builds
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DMVF_PROBE_...] tools/gram_probe.hip -o gram_probe
#include "../spateo-release_amd/csrc/mvf_lib.hip"
#include "../spateo-release_amd/csrc/mvf_gram.hip"
#include <vector>
#include <random>
#include <cstring>
#include <cmath>
namespace mvf {
#include "gram_pc_kernel.h"
// the tile stage of mvf_gram_cached with the producer / consumer kernel (same plan, same partial-tile buffer, same reduction)
static int gram_pc_tiles(const float* ublk, const float* P, int64_t n, int64_t m, double* G, void* workspace) {
    const GramPlan p = make_plan(n, m, MVF_F32);
    if (hipFuncSetAttribute((const void*)gram_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES) != hipSuccess) return 1;
    for (int64_t ph = 0; ph < p.nphases; ++ph) {
        const int64_t s0 = ph * p.phase_slices, ns = std::min(p.phase_slices, p.nslices - s0);
        const unsigned njobs = (unsigned)(ns * p.npairs);
        hipLaunchKernelGGL(gram_pc_kernel, dim3(njobs), dim3(512), PC_LDS_BYTES, 0, ublk, P, n, ublk_npad(n), m, p.nt, p.npairs,
                           p.slice_len, s0, (double*)workspace);
        if (ph + 1 < p.nphases)
            hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, (unsigned)p.npairs), dim3(256), 0, 0,
                               (const double*)workspace, ns, p.nt, p.npairs, m, G, ph > 0 ? 1 : 0);
    }
    return hipGetLastError() != hipSuccess;
}
}  // namespace mvf

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000, m = argc > 2 ? atoll(argv[2]) : 3000;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> hx(n * 4), hc(m * 4), hp(n), hy(n * 4);
    for (int64_t i = 0; i < n; ++i) { hx[4*i] = 2000*u(rng); hx[4*i+1] = 1200*u(rng); hx[4*i+2] = 900*u(rng); hx[4*i+3] = 0; hp[i] = 0.5f + 0.5f*u(rng)*u(rng); hy[4*i]=u(rng); hy[4*i+1]=u(rng); hy[4*i+2]=u(rng); hy[4*i+3]=0; }
    for (int64_t j = 0; j < m; ++j) { hc[4*j] = 2000*u(rng); hc[4*j+1] = 1200*u(rng); hc[4*j+2] = 900*u(rng); hc[4*j+3] = 0; }
    float *x, *c, *p, *y, *ub; double *G, *R; void* ws;
    hipMalloc(&x, n*16); hipMalloc(&c, m*16); hipMalloc(&p, n*4); hipMalloc(&y, n*16);
    hipMemcpy(x, hx.data(), n*16, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), m*16, hipMemcpyHostToDevice);
    hipMemcpy(p, hp.data(), n*4, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), n*16, hipMemcpyHostToDevice);
    hipMalloc(&G, m*m*8); hipMalloc(&R, m*3*8);
    size_t wsb = mvf_gram_workspace_bytes(n, m, MVF_F32); hipMalloc(&ws, wsb);
    size_t ubb = mvf_ublk_bytes(n, m, MVF_F32); hipMalloc(&ub, ubb);
    const double beta = 2.7e-6;
    if (mvf_ublk_build(x, n, c, m, beta, ub, ubb, MVF_F32, nullptr)) { printf("build failed: %s\n", mvf_last_error()); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = (double)n * m * (m + 1);
    auto timeit = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-28s %8.2f ms  %6.1f TF(alg)\n", name, ms, flops / ms / 1e9);
    };
    timeit("recompute f64acc<float>", [&] { mvf_gram_stages(1, x, p, y, n, c, m, beta, G, R, ws, wsb, MVF_F32, nullptr); });
    timeit("cached-U", [&] { mvf_gram_cached(1, ub, x, p, y, n, c, m, beta, G, R, ws, wsb, MVF_F32, nullptr); });
    timeit("ublk_build", [&] { mvf_ublk_build(x, n, c, m, beta, ub, ubb, MVF_F32, nullptr); });
    // A/B: producer / consumer-wave kernel (developer option gram_pc) against the shipped one, G compared bit for bit
    std::vector<double> g0((size_t)m * m), g1((size_t)m * m);
    mvf_gram_cached(1 | 4, ub, x, p, y, n, c, m, beta, G, R, ws, wsb, MVF_F32, nullptr);
    hipMemcpy(g0.data(), G, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    hipMemset(G, 0, (size_t)m * m * 8);
    int rc = mvf::gram_pc_tiles(ub, p, n, m, G, ws);
    if (!rc) rc = mvf_gram_cached(4, ub, x, p, y, n, c, m, beta, G, R, ws, wsb, MVF_F32, nullptr);
    hipError_t he = hipDeviceSynchronize();
    if (rc || he != hipSuccess) { printf("gram_pc failed: rc=%d %s / %s\n", rc, mvf_last_error(), hipGetErrorString(he)); return 1; }
    hipMemcpy(g1.data(), G, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    size_t ndiff = 0; double maxd = 0;
    for (size_t i = 0; i < g0.size(); ++i) if (memcmp(&g0[i], &g1[i], 8)) { ++ndiff; maxd = std::max(maxd, std::fabs(g0[i] - g1[i])); }
    printf("gram_pc vs shipped: %zu of %zu entries differ (max |d| %.3e)\n", ndiff, g0.size(), maxd);
    timeit("cached-U gram_pc", [&] { mvf::gram_pc_tiles(ub, p, n, m, G, ws); });
    timeit("cached-U shipped (again)", [&] { mvf_gram_cached(1, ub, x, p, y, n, c, m, beta, G, R, ws, wsb, MVF_F32, nullptr); });
    return 0;
}
