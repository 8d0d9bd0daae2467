// How one wave issues v_mfma_f64_16x16x4 (diagnostic): accumulators in flight, distinct operands, s_nop / VALU work between MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak/mfma_issue.hip -o tools/mfma_peak/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, bool DISTINCT, int NOPS>
__global__ __launch_bounds__(256) void k_f64(double *out, int iters) {
    d4 acc[NACC];
    double x[NACC], y[NACC];
    for (int a = 0; a < NACC; ++a) { acc[a] = d4{0, 0, 0, 0}; x[a] = threadIdx.x * 1e-3 + a; y[a] = blockIdx.x * 1e-3 - a; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(DISTINCT ? x[a] : x[0], DISTINCT ? y[a] : y[0], acc[a], 0, 0, 0);
            if (NOPS == 1) asm volatile("s_nop 7");
            if (NOPS == 2) { x[a] = x[a] * 1.0000001 + 1e-9; }
        }
    }
    double s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3] + x[a];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}
#define RUN(NACC, D, NOPS, label) { float ms = timeit([&] { hipLaunchKernelGGL((k_f64<NACC, D, NOPS>), dim3(grid), dim3(256), 0, 0, (double *)buf, iters); }); \
    printf("%-44s %d wave/SIMD: %7.2f ms  %6.1f cycles@2.4 per MFMA per SIMD  %.1f TF\n", label, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * w), (double)grid*4*iters*NACC*2048.0/ms/1e9); }
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); const int cus = p.multiProcessorCount;
    void *buf; hipMalloc(&buf, (size_t)cus * 8 * 512 * 8);
    const int iters = 40000;
    for (int w = 1; w <= 2; ++w) {
        const int grid = cus * w;
        RUN(4, false, 0, "4 accs, same operands");
        RUN(8, false, 0, "8 accs, same operands");
        RUN(16, false, 0, "16 accs, same operands");
        RUN(8, true, 0, "8 accs, distinct operands");
        RUN(8, true, 1, "8 accs, distinct, s_nop 7 after each");
        RUN(8, true, 2, "8 accs, distinct, 2 fp64 VALU after each");
    }
    return 0;
}
