// Microbenchmark: what can ONE wave issue in the shadow of v_mfma_f64_16x16x4 (64 cycles)?
// Each iteration: 4 x [1 MFMA + NV filler ops of one kind], fenced so the order is literal.  Diagnostic tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV, int NM, int HALF = 0>
__global__ void k_mix(double* out, long long* cyc, int iters) {
  __shared__ double lds[256 * 4];
  if (HALF && (threadIdx.x & 32)) return;  // only lanes 0..31 of every wave stay active
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  int v[16];
  for (int i = 0; i < 16; ++i) v[i] = i + threadIdx.x;
  double dv[8];
  for (int i = 0; i < 8; ++i) dv[i] = i + threadIdx.x;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  int sacc = iters;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < NM) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = (m * NV + i) & 15;
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dv[j & 7]) : "v"(b), "v"(a));
        if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 15]));
        if (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(v[(j + 1) & 15]));
        if (KIND == 3) lds[threadIdx.x + 256 * (j & 3)] = dv[j & 7];
        if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dv[j & 7]) : "v"(b));
        if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[j]) : "v"(v[(j + 1) & 15]));
        if (KIND == 7) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(dv[j & 7]), "v"(v[j]) : "vcc");
        if (KIND == 8) asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, 3" : "+v"(v[j]) : "s"(sacc));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  double s = sacc;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += dv[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// two waves per SIMD with different roles: waves 0-3 issue only MFMAs, waves 4-7 only VALU (KIND 1 int, 0 fp64)
template <int KIND, int NV, int NM>
__global__ __launch_bounds__(512) void k_roles(double* out, long long* cyc, int iters) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  int v[16];
  for (int i = 0; i < 16; ++i) v[i] = i + threadIdx.x;
  double dv[8];
  for (int i = 0; i < 8; ++i) dv[i] = i + threadIdx.x;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < NM) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4 * NV; ++i) {
        const int j = i & 15;
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dv[j & 7]) : "v"(b), "v"(a));
        if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 15]));
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += dv[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <class F> void run_roles(const char* name, F launch, int iters) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 4096 * 8);
  launch(out, cyc, 10); (void)hipDeviceSynchronize();
  launch(out, cyc, iters); (void)hipDeviceSynchronize();
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-64s MFMA wave %.1f, VALU wave %.1f cycles per iteration\n", name, (double)h[0] / iters, (double)h[4] / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}
template <class F> void run(const char* name, F launch, int iters) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 4096 * 8);
  launch(out, cyc, 10); (void)hipDeviceSynchronize();
  launch(out, cyc, iters); (void)hipDeviceSynchronize();
  long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-64s %.1f cycles per iteration\n", name, (double)h[0] / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  const int it = 20000, G = 256;  // one WG per CU: one wave per SIMD
#define MIX(KIND, NV, NM, WHAT) run("4 x [" #NM ">m MFMA + " #NV " x " WHAT "]", [&](double* o, long long* c, int n) { k_mix<KIND, NV, NM><<<G, 256>>>(o, c, n); }, it)
  MIX(1, 0, 4, "-");
  MIX(0, 4, 4, "v_fma_f64"); MIX(0, 8, 4, "v_fma_f64"); MIX(0, 8, 0, "v_fma_f64");
  MIX(5, 8, 4, "v_add_f64"); MIX(5, 8, 0, "v_add_f64");
  MIX(1, 4, 4, "v_add_u32"); MIX(1, 8, 4, "v_add_u32"); MIX(1, 12, 4, "v_add_u32"); MIX(1, 16, 4, "v_add_u32"); MIX(1, 16, 0, "v_add_u32");
  MIX(2, 8, 4, "v_cndmask_b32"); MIX(2, 16, 4, "v_cndmask_b32"); MIX(2, 16, 0, "v_cndmask_b32");
  MIX(6, 8, 4, "v_mov_b32_dpp"); MIX(6, 8, 0, "v_mov_b32_dpp");
  MIX(7, 8, 4, "v_cmp_class_f64"); MIX(7, 8, 0, "v_cmp_class_f64");
  MIX(8, 8, 4, "s_nop1+v_writelane"); MIX(8, 8, 0, "s_nop1+v_writelane");
  MIX(3, 4, 4, "ds_write_b64"); MIX(3, 8, 4, "ds_write_b64"); MIX(3, 8, 0, "ds_write_b64");
  MIX(4, 8, 4, "s_add_u32"); MIX(4, 16, 4, "s_add_u32"); MIX(4, 16, 0, "s_add_u32");
#define MIXH(KIND, NV, NM, WHAT) run("lanes 0-31 only: 4 x [" #NM ">m MFMA + " #NV " x " WHAT "]", [&](double* o, long long* c, int n) { k_mix<KIND, NV, NM, 1><<<G, 256>>>(o, c, n); }, it)
  MIXH(0, 8, 0, "v_fma_f64"); MIXH(1, 16, 0, "v_add_u32"); MIXH(5, 8, 0, "v_add_f64");
#define ROLES(KIND, NV, NM, WHAT) run_roles("roles: waves0-3 " #NM " MFMA | waves4-7 4x" #NV " " WHAT, [&](double* o, long long* c, int n) { k_roles<KIND, NV, NM><<<G, 512>>>(o, c, n); }, it)
  ROLES(1, 8, 4, "v_add_u32"); ROLES(1, 16, 4, "v_add_u32"); ROLES(1, 16, 0, "v_add_u32"); ROLES(1, 0, 4, "-");
  ROLES(0, 8, 4, "v_fma_f64"); ROLES(0, 8, 0, "v_fma_f64");
  return 0;
}
