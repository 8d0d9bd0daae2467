// Issue rate of v_mfma_f64_4x4x4 against v_mfma_f64_16x16x4 (diagnostic): hipcc --offload-arch=gfx950 -O3 tools/mfma_peak/mfma_4x4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_44(double *out, int iters) {
    double acc[NACC], x[NACC], y[NACC];
    for (int a = 0; a < NACC; ++a) { acc[a] = 0; x[a] = threadIdx.x * 1e-3 + a; y[a] = blockIdx.x * 1e-3 - a; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[a], y[a], acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_16(double *out, int iters) {
    d4 acc[NACC]; double x[NACC], y[NACC];
    for (int a = 0; a < NACC; ++a) { acc[a] = d4{0,0,0,0}; x[a] = threadIdx.x * 1e-3 + a; y[a] = blockIdx.x * 1e-3 - a; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[a], y[a], acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); const int cus = p.multiProcessorCount;
    void *buf; hipMalloc(&buf, (size_t)cus * 8 * 512 * 8);
    const int iters = 100000;
    for (int w = 1; w <= 4; w *= 2) {
        const int grid = cus * w;
        float ms = timeit([&] { hipLaunchKernelGGL((k_44<8>), dim3(grid), dim3(256), 0, 0, (double *)buf, iters); });
        printf("f64 4x4x4 (4 blocks), 8 accs, %d wave/SIMD: %7.2f ms  %6.1f cycles@2.4 per MFMA per SIMD  %.1f TF\n", w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * w), (double)grid*4*iters*8*512.0/ms/1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k_16<8>), dim3(grid), dim3(256), 0, 0, (double *)buf, iters / 4); });
        printf("f64 16x16x4,          8 accs, %d wave/SIMD: %7.2f ms  %6.1f cycles@2.4 per MFMA per SIMD  %.1f TF\n", w, ms, ms * 1e-3 * 2.4e9 / ((double)(iters/4) * 8 * w), (double)grid*4*(iters/4)*8*2048.0/ms/1e9);
    }
    return 0;
}
