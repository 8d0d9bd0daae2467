// v_mfma_f64_4x4x4_4b_f64 on gfx950: issue interval and lane layout (diagnostic).
// Layout probe: A[l] = 1000 + l, B[l] = l for every lane; each D lane is then a sum of 4 products, from which the
// (lane -> block, i, k) map of A and (lane -> block, k, j) map of B are read off on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void k_time(double* out, long long* cyc, int iters) {
  double acc[4] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[m], 0, 0, 0);
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_probe(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 4096 * 8);
  k_time<<<256, 256>>>(out, cyc, 10); (void)hipDeviceSynchronize();
  const int it = 20000;
  k_time<<<256, 256>>>(out, cyc, it); (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("v_mfma_f64_4x4x4_4b: %.1f cycles per instruction (4 independent accumulators, 1 wave/SIMD)\n", (double)h / it / 4);
  // layout: one-hot probes.  For each (la, lb) pair too many; instead use structured values:
  // A[l] = 2^(l%8) * (1 + ...)?  Simpler: for each A lane la set A = e_la (one-hot), B = all (1 + lane/64): D[l] tells
  // which D lanes read A lane la and with which B lane (value identifies the B lane).
  std::vector<double> ha(64), hb(64), hd(64);
  double *da, *db, *dd;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
  for (int l = 0; l < 64; ++l) hb[l] = 100 + l;
  (void)hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
  for (int la = 0; la < 64; ++la) {
    for (int l = 0; l < 64; ++l) ha[l] = (l == la) ? 1.0 : 0.0;
    (void)hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice);
    k_probe<<<1, 64>>>(da, db, dd); (void)hipDeviceSynchronize();
    (void)hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost);
    printf("A lane %2d feeds:", la);
    for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" D%d<-B%d", l, (int)std::lround(hd[l]) - 100);
    printf("\n");
  }
  return 0;
}
