// Developer probe: Gram kernels (cached-U and recompute) for float AND double cell records at several shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gram_probe64.hip -o tools/gram_probe64
#include "../spateo-release_amd/csrc/mvf_lib.hip"
#include "../spateo-release_amd/csrc/mvf_gram.hip"
#include <vector>
#include <random>

template <typename T>
static void run(int64_t n, int64_t m, mvf_dtype dt) {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<T> hx(n * 4), hc(m * 4), hp(n), hy(n * 4);
    for (int64_t i = 0; i < n; ++i) { hx[4*i] = 2000*u(rng); hx[4*i+1] = 1200*u(rng); hx[4*i+2] = 900*u(rng); hx[4*i+3] = 0; hp[i] = 0.5f + 0.5f*u(rng)*u(rng); hy[4*i]=u(rng); hy[4*i+1]=u(rng); hy[4*i+2]=u(rng); hy[4*i+3]=0; }
    for (int64_t j = 0; j < m; ++j) { hc[4*j] = 2000*u(rng); hc[4*j+1] = 1200*u(rng); hc[4*j+2] = 900*u(rng); hc[4*j+3] = 0; }
    T *x, *c, *p, *y; void* ub; double *G, *R; void* ws;
    const size_t s = sizeof(T);
    hipMalloc(&x, n*4*s); hipMalloc(&c, m*4*s); hipMalloc(&p, n*s); hipMalloc(&y, n*4*s);
    hipMemcpy(x, hx.data(), n*4*s, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), m*4*s, hipMemcpyHostToDevice);
    hipMemcpy(p, hp.data(), n*s, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), n*4*s, hipMemcpyHostToDevice);
    hipMalloc(&G, m*m*8); hipMalloc(&R, m*3*8);
    size_t wsb = mvf_gram_workspace_bytes(n, m, dt); hipMalloc(&ws, wsb);
    size_t ubb = mvf_ublk_bytes(n, m, dt); hipMalloc(&ub, ubb);
    const double beta = 2.7e-6;
    if (mvf_ublk_build(x, n, c, m, beta, ub, ubb, dt, nullptr)) { printf("build failed: %s\n", mvf_last_error()); return; }
    hipDeviceSynchronize();
    mvf::GramPlan pl = mvf::make_plan(n, m, dt);
    printf("n=%lld m=%lld %s: pairs=%d slices=%lld slice_len=%lld jobs=%lld (%.2f rounds) ublk=%.1f GB\n", (long long)n, (long long)m,
           dt == MVF_F32 ? "f32" : "f64", pl.npairs, (long long)pl.nslices, (long long)pl.slice_len,
           (long long)(pl.nslices * pl.npairs), pl.nslices * pl.npairs / (double)mvf::gram_slots(), ubb / 1e9);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = (double)n * m * (m + 1);
    auto timeit = [&](const char* name, int stages, bool cached) {
        auto f = [&] { if (cached) mvf_gram_cached(stages, ub, x, p, y, n, c, m, beta, G, R, ws, wsb, dt, nullptr);
                       else mvf_gram_stages(stages, x, p, y, n, c, m, beta, G, R, ws, wsb, dt, nullptr); };
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("   %-24s %8.2f ms  %6.1f TF(alg)\n", name, ms, flops / ms / 1e9);
    };
    {
        hipEventRecord(e0); for (int r = 0; r < 3; ++r) mvf_ublk_build(x, n, c, m, beta, ub, ubb, dt, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("   %-24s %8.2f ms  %6.2f TB/s written\n", "ublk_build", ms, ubb / ms / 1e9);
    }
    timeit("tiles cached", 1, true);
    timeit("tiles recompute", 1, false);
    timeit("rhs", 2, true);
    timeit("reduce", 4, true);
    hipFree(x); hipFree(c); hipFree(p); hipFree(y); hipFree(G); hipFree(R); hipFree(ws); hipFree(ub);
}

int main(int argc, char** argv) {
    for (int a = 1; a + 2 < argc + 0 || a + 2 == argc; a += 3) {
        const int64_t n = atoll(argv[a]), m = atoll(argv[a + 1]);
        if (argv[a + 2][0] == 'd') run<double>(n, m, MVF_F64); else run<float>(n, m, MVF_F32);
    }
    return 0;
}
