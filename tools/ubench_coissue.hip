// Microbenchmark: can ONE wave overlap v_mfma_f64_16x16x4 with f64 VALU work (MFMA : N fma interleave)?
// Also: dependent-chain latency of v_fma_f64.  Diagnostic tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, int NM>
__global__ void k_mix(double* out, long long* cyc, int iters) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = i + threadIdx.x;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < NM) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) v[(m * NV + i) & 15] = fma(v[(m * NV + i) & 15], b, a);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_chain(double* out, long long* cyc, int iters) {
  double v = threadIdx.x, a = 1.0 + threadIdx.x * 1e-9, b = 1e-3;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v = fma(v, a, b);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class F> void run(const char* name, F launch, int iters, double per) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 4096 * 8);
  launch(out, cyc, 10); (void)hipDeviceSynchronize();
  launch(out, cyc, iters); (void)hipDeviceSynchronize();
  long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-52s %.1f cycles per %s\n", name, (double)h[0] / iters / per, per == 1 ? "iteration (4 MFMA slots)" : "fma");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  const int it = 20000, G = 256;  // one WG per CU: one wave per SIMD
#define MIX(NV, NM) run("mix: 4 x [" #NM ">m MFMA + " #NV " fma_f64], 1 wave/SIMD", [&](double* o, long long* c, int n) { k_mix<NV, NM><<<G, 256>>>(o, c, n); }, it, 1)
  MIX(0, 4); MIX(2, 4); MIX(4, 4); MIX(6, 4); MIX(8, 4); MIX(10, 4); MIX(12, 4); MIX(8, 0); MIX(12, 0);
  run("dependent v_fma_f64 chain, 1 wave/SIMD", [&](double* o, long long* c, int n) { k_chain<<<G, 256>>>(o, c, n); }, it, 16);
  return 0;
}
