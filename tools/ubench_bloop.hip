// Microbenchmark of the b = X~ C loop shape: ONE fp64 MFMA accumulator chain, two LDS operands per MFMA,
// 32 k-steps.  Compares coding styles.  Diagnostic tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int VAR>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
  __shared__ double X[32 * 258];
  __shared__ double C[256 * 11];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  for (int i = threadIdx.x; i < 32 * 258; i += 256) X[i] = i * 1e-4;
  for (int i = threadIdx.x; i < 256 * 11; i += 256) C[i] = i * 1e-3;
  __syncthreads();
  const double* xrow = X + (16 * (wave & 1) + l15) * 258 + 128 * (wave >> 1) + l4;
  const double* cpc = C + (128 * (wave >> 1) + l4) * 11 + (l15 < 10 ? l15 : 10);
  d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) {  // naive, fully unrolled
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = MF(xrow[4 * s], cpc[4 * s * 11], acc);
    } else if (VAR == 1) {  // all operands first, then the MFMAs
      double x[32], c[32];
#pragma unroll
      for (int s = 0; s < 32; ++s) { x[s] = xrow[4 * s]; c[s] = cpc[4 * s * 11]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = MF(x[s], c[s], acc);
    } else if (VAR == 2) {  // two independent chains (even / odd k-steps)
#pragma unroll
      for (int s = 0; s < 32; s += 2) { acc = MF(xrow[4 * s], cpc[4 * s * 11], acc); acc2 = MF(xrow[4 * s + 4], cpc[(4 * s + 4) * 11], acc2); }
    } else if (VAR == 3) {  // halves: 16 operands ahead
      double x[16], c[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) { x[s] = xrow[4 * s]; c[s] = cpc[4 * s * 11]; }
      __builtin_amdgcn_sched_barrier(0);
      double x2[16], c2[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) { x2[s] = xrow[4 * (s + 16)]; c2[s] = cpc[4 * (s + 16) * 11]; }
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = MF(x[s], c[s], acc);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = MF(x2[s], c2[s], acc);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[3] + acc2[0] + acc2[2];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int VAR> void run(const char* name) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 4096 * 8);
  k<VAR><<<256, 256>>>(out, cyc, 10); (void)hipDeviceSynchronize();
  const int it = 2000;
  k<VAR><<<256, 256>>>(out, cyc, it); (void)hipDeviceSynchronize();
  long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-50s %.1f cycles per MFMA\n", name, (double)h[0] / it / 32.0);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("naive unrolled, one chain");
  run<1>("all 64 operands loaded first, one chain");
  run<2>("two independent chains");
  run<3>("operands in two halves of 16, one chain");
  return 0;
}
