// Microbenchmark of a P4-like loop: 24 fp64 MFMA accumulator tiles (AGPRs), operands from LDS and from a
// mask word.  Variants isolate what costs more than 64 cycles per MFMA.  Diagnostic tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, unsigned long long mword) {
  __shared__ double W[32 * 82];
  __shared__ double X[32 * 258];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  for (int i = threadIdx.x; i < 32 * 82; i += 256) W[i] = i * 1e-3;
  for (int i = threadIdx.x; i < 32 * 258; i += 256) X[i] = i * 1e-4;
  __syncthreads();
  d4 acc[4][6];
  for (int r = 0; r < 4; ++r) for (int t = 0; t < 6; ++t) acc[r][t] = d4{0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const int smp = 4 * s + l4;
      double bw[5];
      if (VAR >= 1) { for (int t = 0; t < 5; ++t) bw[t] = W[smp * 82 + 16 * t + l15]; }
      else { for (int t = 0; t < 5; ++t) bw[t] = 1.0 + t + lane; }
      const unsigned long long mw = (VAR >= 2) ? (mword >> ((smp + it) & 7)) : mword;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double am = 1.0, ax = 2.0 + r;
        if (VAR >= 2) am = ((mw >> (16 * r + l15)) & 1ull) ? 1.0 : 0.0;
        if (VAR >= 3) ax = X[smp * 258 + 64 * wave + 16 * r + l15];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[r][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bw[t], acc[r][t], 0, 0, 0);
        acc[r][5] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bw[4], acc[r][5], 0, 0, 0);
      }
    }
  }
  long long t1 = clock64();
  double sum = 0;
  for (int r = 0; r < 4; ++r) for (int t = 0; t < 6; ++t) sum += acc[r][t][0] + acc[r][t][3];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int VAR> void run(const char* name) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 4096 * 8);
  k<VAR><<<256, 256>>>(out, cyc, 10, 0x5555AAAA3333CCCCull); (void)hipDeviceSynchronize();
  const int it = 2000;
  k<VAR><<<256, 256>>>(out, cyc, it, 0x5555AAAA3333CCCCull); (void)hipDeviceSynchronize();
  long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-60s %.1f cycles per MFMA\n", name, (double)h[0] / it / 192.0);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("24 acc tiles, operands in registers");
  run<1>("+ B operands from LDS");
  run<2>("+ A operand from mask word (shift/and/cndmask)");
  run<3>("+ x~ operand from LDS (full P4 shape)");
  return 0;
}
