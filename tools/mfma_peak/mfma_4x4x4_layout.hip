// Lane layout of v_mfma_f64_4x4x4, probed with unit operands: prints "A-lane B-lane D-lane" for every non-zero product (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out) {   // out[la*64+lb] = bitmask-lo/hi of lanes whose D is nonzero -> store lane index + 1 (assume at most one)
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = lane + 1;
        }
}
int main() {
    int *d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0, 4096 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) printf("%d %d %d\n", la, lb, h[la * 64 + lb] - 1);
    return 0;
}
