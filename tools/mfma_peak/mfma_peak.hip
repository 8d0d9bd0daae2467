// Achievable MFMA issue rates on this part (diagnostic): back-to-back independent MFMAs, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak/mfma_peak.hip -o tools/mfma_peak/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_i8(int *out, int iters) {
    i4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = i4{0, 0, 0, 0};
    i4 x = {(int)threadIdx.x, 1, 2, 3}, y = {4, 5, (int)blockIdx.x, 7};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, acc[a], 0, 0, 0);
    }
    int s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double *out, int iters) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = d4{0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main(int argc, char **argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    void *buf;
    hipMalloc(&buf, (size_t)cus * 8 * 512 * 8);
    const int iters = 400000;
    double best_i8 = 0, best_f64 = 0;
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
        const int grid = cus * wg_per_cu;
        float ms = timeit([&] { hipLaunchKernelGGL(k_i8<16>, dim3(grid), dim3(256), 0, 0, (int *)buf, iters); });
        double ops = (double)grid * 4 * iters * 16 * 32768.0;
        if (ops / ms / 1e12 > best_i8) best_i8 = ops / ms / 1e12;
        printf("int8 16x16x64, %d wave(s)/SIMD: %.2f ms, %.2f Pop/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", wg_per_cu, ms, ops / ms / 1e12,
               ms * 1e-3 * 2.4e9 / (iters * 16.0 * wg_per_cu));
        ms = timeit([&] { hipLaunchKernelGGL(k_f64<8>, dim3(grid), dim3(256), 0, 0, (double *)buf, iters / 4); });
        ops = (double)grid * 4 * (iters / 4) * 8 * 2048.0;
        if (ops / ms / 1e9 > best_f64) best_f64 = ops / ms / 1e9;
        printf("f64 16x16x4,   %d wave(s)/SIMD: %.2f ms, %.2f Tflop/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", wg_per_cu, ms, ops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((iters / 4) * 8.0 * wg_per_cu));
    }
    // last line: JSON for profiles/rNN/mfma_peak.json (argv[1] = commit stamp)
    printf("{\"fp64_16x16x4_tflops_best\": %.2f, \"int8_16x16x64_pops_best\": %.3f, \"cus\": %d, \"commit\": \"%s\"}\n", best_f64, best_i8, cus,
           argc > 1 ? argv[1] : "");
    return 0;
}
