// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64, v_fma_f64 and i8 MFMA on gfx950
// (cycles per instruction per wave, 1 wave/SIMD and 2 waves/SIMD).  Diagnostic tool, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k_mfma64(double* out, long long* cyc, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k_fma64(double* out, long long* cyc, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k_mfma_i8(int* out, long long* cyc, int iters) {
  i4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i4{0, 0, 0, 0};
  i4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  int s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k_mfma_i8_32(int* out, long long* cyc, int iters) {
  i16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  i4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  int s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F> void run(const char* name, F launch, int insts_per_iter, int iters, int threads) {
  double* out; long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, cyc, 10);  // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[1024]; hipMemcpy(h, cyc, 1024 * 8, hipMemcpyDeviceToHost);
  double c = (double)h[0] / ((double)iters * insts_per_iter);
  printf("%-44s threads/WG %4d: %.1f cycles/inst/wave (WG0), kernel %.3f ms\n", name, threads, c, ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 20000, G = 1024;
  run("v_mfma_f64_16x16x4 x1 acc (dependent)", [&](double* o, long long* c, int n) { k_mfma64<1><<<G, 256>>>(o, c, n); }, 1, it, 256);
  run("v_mfma_f64_16x16x4 x4 acc", [&](double* o, long long* c, int n) { k_mfma64<4><<<G, 256>>>(o, c, n); }, 4, it, 256);
  run("v_mfma_f64_16x16x4 x8 acc", [&](double* o, long long* c, int n) { k_mfma64<8><<<G, 256>>>(o, c, n); }, 8, it, 256);
  run("v_mfma_f64_16x16x4 x8 acc, 2 waves/SIMD", [&](double* o, long long* c, int n) { k_mfma64<8><<<G, 512>>>(o, c, n); }, 8, it, 512);
  run("v_fma_f64 x8 acc", [&](double* o, long long* c, int n) { k_fma64<8><<<G, 256>>>(o, c, n); }, 8, it, 256);
  run("v_fma_f64 x16 acc", [&](double* o, long long* c, int n) { k_fma64<16><<<G, 256>>>(o, c, n); }, 16, it, 256);
  run("v_fma_f64 x16 acc, 2 waves/SIMD", [&](double* o, long long* c, int n) { k_fma64<16><<<G, 512>>>(o, c, n); }, 16, it, 512);
  run("v_fma_f64 x16 acc, 4 waves/SIMD", [&](double* o, long long* c, int n) { k_fma64<16><<<G, 1024>>>(o, c, n); }, 16, it, 1024);
  run("v_mfma_i32_16x16x64_i8 x8 acc", [&](double* o, long long* c, int n) { k_mfma_i8<8><<<G, 256>>>((int*)o, c, n); }, 8, it, 256);
  run("v_mfma_i32_32x32x32_i8 x4 acc", [&](double* o, long long* c, int n) { k_mfma_i8_32<4><<<G, 256>>>((int*)o, c, n); }, 4, it, 256);
  return 0;
}
