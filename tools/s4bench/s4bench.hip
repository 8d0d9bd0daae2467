// Micro-benchmark of solve4_kernel with per-phase cycle counters (diagnostic; built with -DS4_TIMING):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DS4_TIMING -I ppca_rs_amd/csrc tools/s4bench/s4bench.hip -o tools/s4bench/s4bench
//   tools/s4bench/s4bench <k> <n>
#include "../../ppca_rs_amd/csrc/ppca_solve4.hip"

#include <cstdio>
#include <vector>
#include <random>

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 32;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 1000000;
    const int kp = k * (k + 1) / 2;
    std::vector<double> G((size_t)4096 * kp), B((size_t)4096 * (k + 1));
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (int i = 0; i < 4096; ++i) {  // G = A A^T / k, packed lower
        std::vector<double> A(k * k);
        for (auto &v : A) v = nd(rng);
        for (int r = 0; r < k; ++r)
            for (int c = 0; c <= r; ++c) {
                double s = 0;
                for (int t = 0; t < k; ++t) s += A[r * k + t] * A[c * k + t];
                G[(size_t)i * kp + r * (r + 1) / 2 + c] = s / k;
            }
        for (int e = 0; e <= k; ++e) B[(size_t)i * (k + 1) + e] = nd(rng);
    }
    double *dG, *dB, *dxx, *dmc, *dsc, *dmodel;
    hipMalloc(&dG, sizeof(double) * n * kp);
    hipMalloc(&dB, sizeof(double) * n * (k + 1));
    hipMalloc(&dxx, sizeof(double) * n);
    hipMalloc(&dmc, sizeof(double) * n);
    hipMalloc(&dsc, sizeof(double) * n * 4);
    hipMalloc(&dmodel, sizeof(double) * 8);
    const double model[8] = {0.9, 0.81, log(0.9), 0, 0, 0, 0, 0};
    hipMemcpy(dmodel, model, sizeof(model), hipMemcpyHostToDevice);
    std::vector<double> ones(n, 40.0);
    hipMemcpy(dxx, ones.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    hipMemcpy(dmc, ones.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    auto fill = [&] {
        for (int64_t i = 0; i < n; i += 4096) {
            const int64_t m = std::min<int64_t>(4096, n - i);
            hipMemcpy(dG + i * kp, G.data(), sizeof(double) * m * kp, hipMemcpyHostToDevice);
            hipMemcpy(dB + i * (k + 1), B.data(), sizeof(double) * m * (k + 1), hipMemcpyHostToDevice);
        }
    };
    ppca::SolveArgs a{};
    a.G = dG; a.Bz = dB; a.xx = dxx; a.mc = dmc; a.w = nullptr; a.n = n; a.k = k; a.model = dmodel; a.sc = dsc; a.em = 1;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        fill();
#ifdef S4_TIMING
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(ppca::s4_dbg), z, sizeof(z));
#endif
        hipEventRecord(e0, 0);
        hipError_t err = ppca::launch_solve4(a, prop.multiProcessorCount, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("k %d n %lld: %.3f ms (%s)\n", k, (long long)n, ms, hipGetErrorString(err));
#ifdef S4_TIMING
        hipMemcpyFromSymbol(z, HIP_SYMBOL(ppca::s4_dbg), sizeof(z));
        const char *names[] = {"load", "diag", "potrf-mma", "trtri", "lauum", "z", "scalars", "output"};
        const double groups = (double)z[15];
        for (int p = 0; p < 8; ++p) printf("  %-10s %9.0f cycles per group and wave\n", names[p], (double)z[p] / groups);
#endif
    }
    std::vector<double> sc(8);
    hipMemcpy(sc.data(), dsc, sizeof(double) * 8, hipMemcpyDeviceToHost);
    printf("sc[0..3] = %g %g %g %g\n", sc[0], sc[1], sc[2], sc[3]);
    return 0;
}
