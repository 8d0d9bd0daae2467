// ppca_generic.hip -- split pipeline for shapes the fused kernel does not cover
// (d > 256 or k > 10, up to k = 64; BASELINE config 4: d = 1024, k = 64).
//
// Same mathematics as pass_kernel (ppca_kernels.hip) and the same reference items
// (ppca/src/ppca_model.rs:195-208 infer_one, :281-358 M-step sweeps, :142-149 llk), but
// the k x k per-sample state no longer fits registers and the d x k(k+1)/2 statistics no
// longer fit one workgroup, so the pass is cut into dense contractions over sample chunks:
//   Q        = vech(c_j c_j^T)                      (d x k')            qtab_kernel
//   G | b    = Mask . Q  |  X~ . C                  (chunk x (k'+k))    gemm_kernel  (fp64 MFMA)
//   solve    : per sample, one wave: Cholesky in LDS, z, M^-1 -> w P, w z, llk   solve_kernel
//   S, U, totals += Mask^T . [wP | wz | w]          (d x (k'+k+1))      gemm_kernel
//   cross, sumx  += X~^T  . [wz | w]                (d x (k+1))         gemm_kernel
// Chunk results are accumulated in chunk order (deterministic).  Correctness first: the
// GEMM is a plain LDS-staged 64x64 tile kernel on v_mfma_f64_16x16x4_f64.
#include <algorithm>
#include <vector>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ppca_internal.hpp"
#include "ppca_solve.hpp"

namespace ppca {

typedef double d4g_t __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_g_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for_g(F &&f) {
    static_for_g_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ double gwave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ Q table
__global__ void qtab_kernel(const double *model, int d, int k, double *q) {
    const int kp = k * (k + 1) / 2;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)d * kp) return;
    int j = (int)(idx / kp), e = (int)(idx - (int64_t)j * kp);
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= e) ++a;
    int b = e - a * (a + 1) / 2;
    const double *c = model + MODEL_HDR + (int64_t)j * k;
    q[idx] = c[a] * c[b];
}

// ------------------------------------------------------------------ row statistics
// xx_i = sum_obs (x - mu)^2, m_i = #observed; one wave per row.
__global__ void rowstats_kernel(const double *X, int64_t ldx, int64_t n, int d, const double *model, int k, double *xx,
                                double *mcount) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const double *mean = model + MODEL_HDR + (int64_t)d * k;
    double s = 0.0, m = 0.0;
    for (int j = lane; j < d; j += 64) {
        double v = X[row * ldx + j];
        if (__builtin_isfinite(v)) {
            double t = v - mean[j];
            s += t * t;
            m += 1.0;
        }
    }
    s = gwave_sum(s);
    m = gwave_sum(m);
    if (lane == 0) {
        xx[row] = s;
        mcount[row] = m;
    }
}

// ------------------------------------------------------------------ GEMM
// C[M x N] (+)= A[M x K] . B[K x N], A generated from the sample matrix:
//   AMODE 0: A[i][j] = mask(X[i][j])            (M = samples, K = dims)
//   AMODE 1: A[i][j] = x~_ij                    (M = samples, K = dims)
//   AMODE 2: A[j][i] = mask(X[i][j])            (M = dims,    K = samples)
//   AMODE 3: A[j][i] = x~_ij                    (M = dims,    K = samples)
// B is dense row-major (ldb).  Output element (r, c): c < ncols0 -> out0[r*ld0 + c],
// else out1[r*ld1 + (c - ncols0)].
struct GemmArgs {
    const double *X;
    int64_t ldx;
    const double *mean;
    const double *B;
    int64_t ldb;
    int64_t M, N, K;
    double *out0;
    int64_t ld0;
    int64_t ncols0;
    double *out1;
    int64_t ld1;
    int accumulate;
    // split-K: blockIdx.z handles K-range [z * kslice, (z+1) * kslice) and writes a dense M x N partial at
    // part + z * M * N (then splitk_reduce_kernel adds the partials to the outputs in slice order)
    int64_t kslice;
    double *part;
    // device-side selection between this fp64 contraction and its int8-sliced twin: run only if *guard == run_if
    const int *guard;
    int run_if;
};

template <int AMODE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ double As[64][17];
    __shared__ double Bs[16][65];
    if (g.guard && *g.guard != g.run_if) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    d4g_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = d4g_t{0, 0, 0, 0};
    const int64_t kbeg = g.part ? (int64_t)blockIdx.z * g.kslice : 0;
    const int64_t kend = g.part ? ((kbeg + g.kslice < g.K) ? kbeg + g.kslice : g.K) : g.K;
    // The operands of K-chunk c+1 are fetched from global memory into registers before the MFMAs of chunk c and go
    // to LDS after them: the global latency hides behind the 1024 MFMA cycles of a chunk instead of sitting between
    // two barriers.
    double ra[4], rb[4];
    auto fetch = [&](int64_t k0) {
        if (AMODE < 2) {  // A tile: 64 (M) x 16 (K); thread -> row tid / 4, four consecutive K
            const int i = tid >> 2, kk = (tid & 3) * 4;
            const int64_t row = m0 + i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t col = k0 + kk + u;
                double v = 0.0;
                if (row < g.M && col < kend) {
                    const double x = g.X[row * g.ldx + col];
                    const bool fin = __builtin_isfinite(x);
                    v = (AMODE == 0) ? (fin ? 1.0 : 0.0) : (fin ? x - g.mean[col] : 0.0);
                }
                ra[u] = v;
            }
        } else {  // transposed: thread -> sample tid / 16 of the K-chunk, four consecutive dims
            const int i = tid >> 4, jj = (tid & 15) * 4;
            const int64_t srow = k0 + i;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t dim = m0 + jj + u;
                double v = 0.0;
                if (srow < kend && dim < g.M) {
                    const double x = g.X[srow * g.ldx + dim];
                    const bool fin = __builtin_isfinite(x);
                    v = (AMODE == 2) ? (fin ? 1.0 : 0.0) : (fin ? x - g.mean[dim] : 0.0);
                }
                ra[u] = v;
            }
        }
        {  // B tile: 16 (K) x 64 (N)
            const int kk = tid >> 4, cc = (tid & 15) * 4;
            const int64_t kr = k0 + kk;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t col = n0 + cc + u;
                rb[u] = (kr < kend && col < g.N) ? g.B[kr * g.ldb + col] : 0.0;
            }
        }
    };
    auto stash = [&]() {
        if (AMODE < 2) {
            const int i = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[i][kk + u] = ra[u];
        } else {
            const int i = tid >> 4, jj = (tid & 15) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[jj + u][i] = ra[u];
        }
        const int kk = tid >> 4, cc = (tid & 15) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) Bs[kk][cc + u] = rb[u];
    };
    if (kbeg < kend) fetch(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += 16) {
        stash();
        __syncthreads();
        if (k0 + 16 < kend) fetch(k0 + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double a = As[16 * wave + l15][4 * s + l4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[4 * s + l4][16 * t + l15], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + 16 * wave + l4 + 4 * r, col = n0 + 16 * t + l15;
            if (row < g.M && col < g.N) {
                if (g.part) {
                    g.part[((int64_t)blockIdx.z * g.M + row) * g.N + col] = acc[t][r];
                } else {
                    double *dst = (col < g.ncols0) ? g.out0 + row * g.ld0 + col : g.out1 + row * g.ld1 + (col - g.ncols0);
                    *dst = g.accumulate ? *dst + acc[t][r] : acc[t][r];
                }
            }
        }
}


// ------------------------------------------------------------------ the two skinny statistics products in ONE pass over X
//     [U | totals] (d x (k+1)) += Mask^T . [wz | w]        (for total_deviation, ppca_model.rs:338-348)
//     [cross | sumx] (d x (k+1)) += X~^T . [wz | w]        (:281-293)
// gemm_kernel<2> / <3> compute them one after the other on 64-column tiles: at k + 1 <= 16 three quarters of every MFMA
// are padding and X is read twice (1.8 + 2.0 ms of a 8 ms chunk at d = 256, k = 11).  Here a wave owns a strip of 64
// dimensions x 16 NT columns of BOTH products; the A operands come straight from global memory (lane = dimension 16 r +
// lane % 16 of sample 4 s + lane / 16: whole 128-byte segments of four rows, no LDS), one read of X feeds both
// products, and the mask / centring is a compare and a select on the way.  Samples are cut into slices (grid.x), the
// slice partials summed in slice order (deterministic).
typedef double sd4_t __attribute__((ext_vector_type(4)));
// NT = 16-column tiles of [wz | w], RT = 16-dimension row tiles per wave: a workgroup covers 256 dimensions with
// 16 / RT waves (RT = 4: four waves, k + 1 <= 32; RT = 2: eight waves, k + 1 <= 80 -- 2 x RT x NT accumulator tiles).
template <int NT, int RT>
__global__ __launch_bounds__(64 * (16 / RT)) void skinny_xt_kernel(const double *X, int64_t ldx, int64_t n, int d, const double *mean,
                                                                 const double *Bz, int ncols, int64_t rows_per_slice, double *part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int dbase = blockIdx.y * 256 + 16 * RT * wave;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slice, r1 = r0 + rows_per_slice < n ? r0 + rows_per_slice : n;
    sd4_t accM[RT][NT], accX[RT][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) accM[r][t] = accX[r][t] = sd4_t{0, 0, 0, 0};
    // Every load is unconditional (clamped to a real element) and the validity applied by selects afterwards: a load
    // inside a divergent branch gets its own exec-masked region and wait, and the loop turns latency-bound.
    double mu[RT];
    bool dok[RT];
    int coff[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int j = dbase + 16 * r + l15;
        dok[r] = j < d;
        coff[r] = dok[r] ? j : d - 1;
        mu[r] = mean[coff[r]];
    }
    int boff[NT];
    bool bok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = 16 * t + l15;
        bok[t] = c < ncols;
        boff[t] = bok[t] ? c : ncols - 1;
    }
    constexpr int UN = 4;  // k-steps (of four samples) whose operands are requested before the first of them is used
    for (int64_t s0 = r0; s0 < r1; s0 += 4 * UN) {
        double xv[UN][RT], bv[UN][NT];
        bool valid[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t row = s0 + 4 * u + l4;
            valid[u] = row < r1;
            const int64_t rc = valid[u] ? row : r1 - 1;
#pragma unroll
            for (int r = 0; r < RT; ++r) xv[u][r] = X[rc * ldx + coff[r]];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[u][t] = Bz[rc * ncols + boff[t]];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            double am[RT], xt[RT], bz[NT];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const bool ob = valid[u] && dok[r] && __builtin_fabs(xv[u][r]) < __builtin_inf();
                am[r] = ob ? 1.0 : 0.0;
                xt[r] = ob ? xv[u][r] - mu[r] : 0.0;  // select, never multiply (utils.rs:118-127)
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) bz[t] = (valid[u] && bok[t]) ? bv[u][t] : 0.0;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    accM[r][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[r], bz[t], accM[r][t], 0, 0, 0);
                    accX[r][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[r], bz[t], accX[r][t], 0, 0, 0);
                }
        }
    }
    // part[slice][product][dim (padded to 256 per grid.y)][16 NT]; C/D map: row = l4 + 4 q, column = l15
    const int dpad = gridDim.y * 256;
    double *pm = part + ((int64_t)blockIdx.x * 2) * dpad * (16 * NT), *px = pm + (int64_t)dpad * (16 * NT);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t dim = dbase + 16 * r + l4 + 4 * q;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pm[dim * (16 * NT) + 16 * t + l15] = accM[r][t][q];
                px[dim * (16 * NT) + 16 * t + l15] = accX[r][t][q];
            }
        }
}
// stats += sum over the slices, in slice order: product 0 -> [U | totals], product 1 -> [cross | sumx]
__global__ void skinny_xt_reduce_kernel(const double *part, int nslices, int dpad, int ncolpad, int d, int k, double *U, double *totals,
                                        double *cross, double *sumx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)dpad * ncolpad;
    if (idx >= 2 * per) return;
    const int prod = (int)(idx / per);
    const int64_t rem = idx - prod * per;
    const int dim = (int)(rem / ncolpad), c = (int)(rem - (int64_t)dim * ncolpad);
    if (dim >= d || c > k) return;
    double v = 0.0;
    for (int sl = 0; sl < nslices; ++sl) v += part[((int64_t)sl * 2 + prod) * per + rem];
    double *dst = prod == 0 ? (c < k ? U + (int64_t)dim * k + c : totals + dim) : (c < k ? cross + (int64_t)dim * k + c : sumx + dim);
    *dst += v;
}
// returns false when the shape is not the skinny kernel's (k + 1 > 80 columns or not enough scratch): the caller falls
// back on gemm_kernel<2> / <3>
static bool launch_skinny_xt(const double *X, int64_t ldx, int64_t n, int d, int k, const double *mean, const double *Bz,
                             double *stats, const StatsLayout &L, double *part_ws, int64_t part_cap, int n_cu, hipStream_t s,
                             hipError_t *err) {
    const int ncols = k + 1, nt = (ncols + 15) / 16;
    if (nt > 5 || n < 1) return false;
    const int gy = (d + 255) / 256, dpad = gy * 256, ncolpad = 16 * nt;
    int64_t slices = std::max<int64_t>(1, (4 * (int64_t)n_cu) / gy);   // four workgroups per CU: the loads of one cover the MFMAs of another
    slices = std::min<int64_t>(slices, (n + 255) / 256);
    slices = std::min<int64_t>(slices, part_cap / (2 * (int64_t)dpad * ncolpad));
    if (slices < 1) return false;
    const int64_t rps = ((n + slices - 1) / slices + 3) / 4 * 4;
    slices = (n + rps - 1) / rps;
    dim3 grid((unsigned)slices, (unsigned)gy);
    switch (nt) {
        case 1: hipLaunchKernelGGL((skinny_xt_kernel<1, 4>), grid, dim3(256), 0, s, X, ldx, n, d, mean, Bz, ncols, rps, part_ws); break;
        case 2: hipLaunchKernelGGL((skinny_xt_kernel<2, 4>), grid, dim3(256), 0, s, X, ldx, n, d, mean, Bz, ncols, rps, part_ws); break;
        case 3: hipLaunchKernelGGL((skinny_xt_kernel<3, 2>), grid, dim3(512), 0, s, X, ldx, n, d, mean, Bz, ncols, rps, part_ws); break;
        case 4: hipLaunchKernelGGL((skinny_xt_kernel<4, 2>), grid, dim3(512), 0, s, X, ldx, n, d, mean, Bz, ncols, rps, part_ws); break;
        default: hipLaunchKernelGGL((skinny_xt_kernel<5, 2>), grid, dim3(512), 0, s, X, ldx, n, d, mean, Bz, ncols, rps, part_ws); break;
    }
    const int64_t tot = 2 * (int64_t)dpad * ncolpad;
    hipLaunchKernelGGL(skinny_xt_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, part_ws, (int)slices, dpad, ncolpad, d, k,
                       stats + L.U, stats + L.totals, stats + L.cross, stats + L.sumx);
    *err = hipGetLastError();
    return true;
}

// ------------------------------------------------------------------ int8-sliced exact contractions
// Both large contractions of the pass have one EXACT operand, the 0/1 mask:
//     G[i][c] = sum_j m_ij Q[j][c]        (samples x k', over the d dims)
//     S[j][c] = sum_i m_ij (wP)[i][c]     (d x k', over the samples of a chunk)
// so, as in the fused kernel (ppca_kernels.hip, qprep_kernel), the other operand is cut into QS = 8 signed 7-bit
// digits of a per-column fixed-point form (54 bits below the column maximum) and contracted on
// v_mfma_i32_16x16x64_i8 with exact integer accumulation -- 8 slices x 1/64 of the fp64 MFMA's cycles per
// multiply-add.  The digit sums are folded into fp64 once per output element (Horner).  Dynamic-range guards decide on
// the DEVICE, per model (Gram) and per chunk (S), whether the int8 form or the fp64-MFMA GEMM runs (both are
// enqueued; the loser returns at once):
//   Gram: the fused kernel's rule (forward bound K d eps <= 1e-8 sigma^2 or backward bound K^2 eps <= 2^-40 r_min)
//   S:    column maximum of |wP| over the chunk <= 2^20 x the column's mean |wP| -- then the absolute error of a
//         sum, <= n 2^-55 max, stays below 2^-34 of the sum of the |terms|.
constexpr int GQS = 8;
typedef int gi4_t __attribute__((ext_vector_type(4)));

// mask bytes of a chunk in both orientations: A[i][dpad] (row = sample) and AT[j][npad] (row = dim), zero padded.
// 64 samples x 64 dims per workgroup: every wave reads 16 rows as whole 512-byte segments (lane = dim), each lane
// drops its flag byte into an LDS tile, and the tile leaves as 16-byte pieces in both orientations.
__global__ __launch_bounds__(256) void gen_maskbytes_kernel(const double *X, int64_t ldx, int64_t n, int d, int dpad,
                                                            int64_t npad, unsigned char *A, unsigned char *AT) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[64][80];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i0 = (int64_t)blockIdx.y * 64;
    const int j0 = blockIdx.x * 64;
    const int j = j0 + lane;
    double v[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int64_t row = i0 + 16 * wave + rr;
        v[rr] = (row < n && j < d) ? X[row * ldx + j] : __builtin_nan("");
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) tile[16 * wave + rr][lane] = __builtin_isfinite(v[rr]) ? 1 : 0;
    __syncthreads();
    const int r = t >> 2, q = t & 3;
    union { unsigned char b[16]; gi4_t vv; } u;
    u.vv = *reinterpret_cast<const gi4_t *>(&tile[r][16 * q]);
    *reinterpret_cast<gi4_t *>(A + (i0 + r) * dpad + j0 + 16 * q) = u.vv;  // rows up to npad: zeros beyond n
#pragma unroll
    for (int e = 0; e < 16; ++e) u.b[e] = tile[16 * q + e][r];
    *reinterpret_cast<gi4_t *>(AT + (int64_t)(j0 + r) * npad + i0 + 16 * q) = u.vv;  // dims up to dpad: zeros beyond d
}

// Everything the split pipeline needs from X before the per-sample solve, in ONE pass over the chunk's rows (rowstats_kernel,
// gen_maskbytes_kernel and gemm_kernel<1> read X once each: three of the pipeline's four passes over X, and X is the
// largest operand of every shape this pipeline serves):
//   xx_i = sum_obs (x - mu)^2, m_i = #observed          (rowstats_kernel's summation order: lane-strided, then the wave)
//   mask bytes in both orientations (when want_bytes)    (gen_maskbytes_kernel's tiles)
//   b = X~ C into Bz[i][0 .. k) (row stride k + 1)       (fp64 MFMA, k <= 16 NT columns)
// One workgroup per 64 samples, looping over 64-dim blocks: every wave reads its 16 rows as whole 512-byte segments
// (lane = dim); x~ and the block's rows of C go through LDS to the MFMA (A: this wave's own 16 rows; row stride 68
// doubles: 8 banks per row, two passes per read, the minimum for 64 doubles).
static size_t prep_lds(int nt) {  // xt [64][68] + cs [64][16 nt + 1] (+ 8 bytes to 16-byte alignment) + tile [64][80]
    const size_t cs2 = 16 * (size_t)nt + 1;
    return sizeof(double) * (64 * 68 + 64 * cs2 + ((64 * cs2) & 1)) + 64 * 80;
}
template <int NT>
__global__ __launch_bounds__(256) void gen_prep_kernel(const double *X, int64_t ldx, int64_t n, int d, int dpad, int64_t npad,
                                                       const double *model, int k, unsigned char *A, unsigned char *AT,
                                                       double *xx, double *mcount, double *Bz, int want_bytes) {
    // (round 6) all NT column tiles of the block's rows of C in LDS at once (k = 64: 33 KB; the whole image 73 KB of dynamic LDS, still
    // two workgroups per CU): with two tiles at a time (static LDS under 64 KB, rounds 4-5) every 64-dimension block paid a second
    // staging of C with a barrier on either side of it.
    constexpr int CT = NT;
    constexpr int XS2 = 68, CS2 = 16 * CT + 1;
    extern __shared__ __attribute__((aligned(16))) double prep_sm[];
    double *xt = prep_sm;                          // [64][XS2]
    double *cs = xt + 64 * XS2;                    // [64][CS2]
    unsigned char (*tile)[80] = reinterpret_cast<unsigned char (*)[80]>(cs + 64 * CS2 + (64 * CS2 & 1));  // [64][80], 16-byte aligned
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const double *mean = model + MODEL_HDR + (int64_t)d * k;
    const double *Cm = model + MODEL_HDR;
    double sxx[16], smc[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) sxx[rr] = smc[rr] = 0.0;
    d4g_t acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = d4g_t{0, 0, 0, 0};
    // the rows of dimension block j0 + 64 are requested as soon as those of block j0 sit in LDS (round 4: the loop paid one
    // global-memory latency per block with nothing of its own to cover it)
    double v[16];
    auto load_block = [&](int jb) {
        const int j = jb + lane;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int64_t row = i0 + 16 * wave + rr;
            v[rr] = (row < n && j < d) ? X[row * ldx + j] : __builtin_nan("");
        }
    };
    load_block(0);
    for (int j0 = 0; j0 < dpad; j0 += 64) {
        const int j = j0 + lane;
        const double mu = j < d ? mean[j] : 0.0;
        auto load_c = [&](int h) {  // columns 16 CT h .. of the block's rows of C (zero past d and past k)
            for (int idx = t; idx < 64 * 16 * CT; idx += 256) {
                const int jj = idx / (16 * CT), a = 16 * CT * h + idx - jj * (16 * CT);
                cs[jj * CS2 + (a - 16 * CT * h)] = (j0 + jj < d && a < k) ? Cm[(int64_t)(j0 + jj) * k + a] : 0.0;
            }
        };
        load_c(0);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const bool fin = __builtin_isfinite(v[rr]);
            const double xv = fin ? v[rr] - mu : 0.0;
            tile[16 * wave + rr][lane] = fin ? 1 : 0;
            xt[(16 * wave + rr) * XS2 + lane] = xv;
            sxx[rr] += xv * xv;
            smc[rr] += fin ? 1.0 : 0.0;
        }
        __syncthreads();
        if (j0 + 64 < dpad) load_block(j0 + 64);
        if (want_bytes) {
            const int r = t >> 2, q = t & 3;
            union { unsigned char b[16]; gi4_t vv; } u;
            u.vv = *reinterpret_cast<const gi4_t *>(&tile[r][16 * q]);
            *reinterpret_cast<gi4_t *>(A + (i0 + r) * dpad + j0 + 16 * q) = u.vv;  // rows up to npad: zeros beyond n
#pragma unroll
            for (int e = 0; e < 16; ++e) u.b[e] = tile[16 * q + e][r];
            *reinterpret_cast<gi4_t *>(AT + (int64_t)(j0 + r) * npad + i0 + 16 * q) = u.vv;  // dims up to dpad: zeros beyond d
        }
#pragma unroll
        for (int h = 0; h < NT / CT; ++h) {
            if (h > 0) {
                __syncthreads();
                load_c(h);
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const double a = xt[(16 * wave + l15) * XS2 + 4 * s + l4];
#pragma unroll
                for (int tt = 0; tt < CT; ++tt)
                    acc[CT * h + tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, cs[(4 * s + l4) * CS2 + 16 * tt + l15], acc[CT * h + tt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const double sx = gwave_sum(sxx[rr]), sm = gwave_sum(smc[rr]);
        const int64_t row = i0 + 16 * wave + rr;
        if (lane == 0 && row < n) {
            xx[row] = sx;
            mcount[row] = sm;
        }
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = i0 + 16 * wave + l4 + 4 * r;
            const int col = 16 * tt + l15;
            if (row < n && col < k) Bz[row * (k + 1) + col] = acc[tt][r];
        }
}

// smallest non-zero squared row norm of C (guard of the Gram digits); one workgroup
__global__ __launch_bounds__(256) void gen_rmin_kernel(const double *model, int d, int k, double *rmin_out, int *flags) {
    __shared__ unsigned long long best;
    if (threadIdx.x == 0) best = 0x7FF0000000000000ull;
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 256) {
        double rn = 0.0;
        for (int a = 0; a < k; ++a) {
            const double c = model[MODEL_HDR + (int64_t)j * k + a];
            rn += c * c;
        }
        if (rn > 0.0 && rn < 1.0e300) atomicMin(&best, (unsigned long long)__double_as_longlong(rn));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *rmin_out = __longlong_as_double((long long)best);
        flags[0] = 0;
    }
}

// digits of Q = vech(c c^T): one workgroup per packed column c; planes BtQ[s][c][dpad], scale[c]
__global__ __launch_bounds__(256) void gen_qdigits_kernel(const double *model, int d, int k, int dpad, int kp,
                                                          const double *rmin, double *scale, signed char *BtQ, int *flags) {
    __shared__ unsigned long long cmax;
    __shared__ int bad;
    const int c = blockIdx.x, t = threadIdx.x;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= c) ++a;
    const int b = c - a * (a + 1) / 2;
    if (t == 0) {
        cmax = 0ull;
        bad = 0;
    }
    __syncthreads();
    double mx = 0.0;
    int notfin = 0;
    for (int j = t; j < d; j += 256) {
        const double q = fabs(model[MODEL_HDR + (int64_t)j * k + a] * model[MODEL_HDR + (int64_t)j * k + b]);
        if (!(q < 1.0e300)) notfin = 1;
        else mx = fmax(mx, q);
    }
    if (mx > 0.0) atomicMax(&cmax, (unsigned long long)__double_as_longlong(mx));
    if (notfin) atomicOr(&bad, 1);
    __syncthreads();
    const double cm = __longlong_as_double((long long)cmax);
    int e = 0;
    if (cm > 0.0) (void)frexp(cm, &e);
    const double sc = ldexp(1.0, e - (7 * GQS - 2));
    if (t == 0) {
        scale[c] = sc;
        int unsafe = bad;
        if (cm > 0.0) {
            const double eps = 0.5 * sc, s2 = model[1];
            const bool fwd = (double)k * (double)d * eps <= 1.0e-8 * s2;
            const bool bwd = (double)k * (double)k * eps <= 9.094947017729282e-13 * *rmin;
            if (!(fwd || bwd)) unsafe = 1;
        }
        if (unsafe) atomicOr(&flags[0], 1);
    }
    const int shift = (7 * GQS - 2) - e;
    for (int j = t; j < dpad; j += 256) {
        double q = 0.0;
        if (j < d) q = model[MODEL_HDR + (int64_t)j * k + a] * model[MODEL_HDR + (int64_t)j * k + b];
        if (!(fabs(q) < 1.0e300)) q = 0.0;
        long long I = llrint(ldexp(q, shift));
#pragma unroll
        for (int sl = 0; sl < GQS; ++sl) {
            const int dig = (int)((I + 64) & 127) - 64;
            I = (I - dig) >> 7;
            BtQ[((int64_t)sl * kp + c) * dpad + j] = (signed char)dig;
        }
    }
}

// per column of V[n][kp]: maximum and sum of |v| over a block of 256 rows -> part[block][2][kp]
__global__ __launch_bounds__(256) void gen_colstat_kernel(const double *V, int64_t n, int kp, double *part, int *flags) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) flags[1] = 0;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= kp) return;
    const int64_t r0 = (int64_t)blockIdx.y * 256, r1 = (r0 + 256 < n) ? r0 + 256 : n;
    double mx = 0.0, sm = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        const double v = fabs(V[r * kp + c]);
        mx = fmax(mx, v);   // (a NaN leaves the maximum alone; the sum below carries it into the guard)
        sm += v;
    }
    part[((int64_t)blockIdx.y * 2) * kp + c] = mx;
    part[((int64_t)blockIdx.y * 2 + 1) * kp + c] = sm;
}
// -> scale[c] and the chunk's guard flag (flags[1], reset by gen_colstat_kernel); sums in block order (deterministic)
// (round 6) pred / redo_col: the chunk's digits were already cut by gen_wdigits_pred_kernel under pred[c] (use_pred = 1).  The column
// keeps that scale if it is the power of two the cut derived from it and lies within [1, 4] x the scale of the column's own maximum
// (no overflow of the 8 x 7-bit form, at most two bits coarser); otherwise the column is marked in redo_col (and flags[2] raised) and
// gen_wdigits_kernel cuts it again under its own scale.  scale[c] = the scale the digits in BtW carry either way; pred[c] for the
// NEXT chunk = twice the scale of this chunk's maximum (one bit of room upwards).
__global__ __launch_bounds__(256) void gen_colscale_kernel(const double *part, int nblocks, int64_t n, int kp, double *scale,
                                                           int *flags, double *pred, int *redo_col, int use_pred) {
    // one workgroup per column: thread t takes blocks t, t + 256, ..., then a fixed tree over the threads
    // (deterministic; one THREAD per column walked a million-row chunk's 4096 block partials alone: 1 ms per chunk)
    __shared__ double rmx[256], rsm[256], rmn[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double mx = 0.0, sm = 0.0, mn = __builtin_inf();
    for (int b = t; b < nblocks; b += 256) {
        const double bs = part[((int64_t)b * 2 + 1) * kp + c];
        const int64_t rows = (int64_t)(b + 1) * 256 <= n ? 256 : n - (int64_t)b * 256;
        mx = fmax(mx, part[((int64_t)b * 2) * kp + c]);
        sm += bs;
        // the smallest mean magnitude of a 256-row block.  A block whose column sum is exactly 0 (all its rows carry weight 0:
        // a run of zero weights, sorted data) holds only zeros, which any cut represents exactly: it cannot be damaged by a
        // coarse scale and must not send the whole chunk to the fp64 engine (advisor, round 4)
        if (bs > 0.0) mn = fmin(mn, bs / (double)rows);
    }
    rmx[t] = mx;
    rsm[t] = sm;
    rmn[t] = mn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            rmx[t] = fmax(rmx[t], rmx[t + o]);
            rsm[t] += rsm[t + o];
            rmn[t] = fmin(rmn[t], rmn[t + o]);
        }
        __syncthreads();
    }
    if (t == 0) {
        mx = rmx[0];
        sm = rsm[0];
        mn = rmn[0];
        int e = 0;
        if (mx > 0.0 && mx < 1.0e300) (void)frexp(mx, &e);
        const double own = ldexp(1.0, e - (7 * GQS - 2));
        double used = own;
        int redo = use_pred;
        if (use_pred) {
            const double ps = pred[c];
            if (ps > 0.0 && ps < 1.0e300) {
                int pe = 0;
                (void)frexp(ps, &pe);
                const double pc = ldexp(1.0, pe - 1);  // (what gen_wdigits_pred_kernel made of it)
                if (pc == ps && own <= pc && pc <= 4.0 * own) {
                    used = pc;
                    redo = 0;
                }
            }
        }
        scale[c] = used;
        if (redo_col) {
            redo_col[c] = redo;
            if (redo) {
                atomicOr(&flags[2], 1);
                atomicAdd(&flags[3], 1);  // (diagnostic count, PPCA_GEN_WPRED_STATS)
            }
        }
        if (pred) pred[c] = 2.0 * own;
        // finite, and the maximum within 2^20 of the mean magnitude of EVERY 256-row block (round 4: the mean over the whole
        // chunk is dominated by the very row that breaks the form -- one row at 1e6 x the others passed "max <= 2^20 x mean"
        // whenever the chunk had fewer than 2^20 rows, and the dimensions masked in that row summed rows cut at 14 bits);
        // under a predicted (coarser) scale the bound tightens by the same factor: the same absolute precision is guaranteed
        if (!(sm < 1.0e300) || !(mx * (used / own) <= 1048576.0 * mn)) atomicOr(&flags[1], 1);
    }
}

// digits of V[n][kp] (row-major) -> planes BtW[s][c][npad] (samples contiguous): 64 samples x 64 columns per
// workgroup, lane = column (rows read as whole 512-byte segments), wave w = samples 16 w .. 16 w + 15 -- the 16 digits
// of one (slice, column) are 16 consecutive bytes of the output: one 16-byte store, no transposition
// redo_col != nullptr: only the columns marked there (flags[2] = any), see gen_colscale_kernel.  Grid (ceil(kp / 64), any): a workgroup
// walks the 64-row tiles blockIdx.y, blockIdx.y + gridDim.y, ...
__global__ __launch_bounds__(256) void gen_wdigits_kernel(const double *V, int64_t n, int64_t ncpad, int kp, int64_t npad,
                                                          const double *scale, signed char *BtW, const int *flags, const int *redo_col) {
    if (flags[1]) return;
    if (redo_col && !flags[2]) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool mine = c < kp && (!redo_col || redo_col[c] != 0);
    if (__ballot(mine) == 0ull) return;
    int ex = 0;
    if (mine) (void)frexp(scale[c], &ex);  // scale = 2^(ex - 1)
    const int shift = -(ex - 1);
    for (int64_t i0 = (int64_t)blockIdx.y * 64 + 16 * wave; i0 < ncpad; i0 += (int64_t)gridDim.y * 64) {
        double v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (i0 + e < n && mine) ? V[(i0 + e) * kp + c] : 0.0;
        if (!mine) continue;
        union { signed char b[GQS][16]; gi4_t q[GQS]; } u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            long long I = llrint(ldexp(v[e], shift));
#pragma unroll
            for (int sl = 0; sl < GQS; ++sl) {
                const int dig = (int)((I + 64) & 127) - 64;
                I = (I - dig) >> 7;
                u.b[sl][e] = (signed char)dig;
            }
        }
#pragma unroll
        for (int sl = 0; sl < GQS; ++sl) *reinterpret_cast<gi4_t *>(BtW + ((int64_t)sl * kp + c) * npad + i0) = u.q[sl];
    }
}

// (round 6) The digit planes of a chunk's wP in full 128-byte lines, and -- STATS -- gen_colstat_kernel's column statistics from the
// same registers.
//   Lane layout: a wave = 8 row groups x 8 columns (lane = 8 rg + cc): lane (rg, cc) holds rows 16 rg .. 16 rg + 15 of a 128-row tile
//   for column c0 + cc, so the 16-byte stores of the eight lanes of a column make one whole line of BtW[slice][column][sample].
//   gen_wdigits_kernel's layout (lane = column, 16 rows per wave) writes 16 bytes of each of 64 lines per store instruction: by
//   ablation (profiles/r06/wdigits_ablation.log) 770 of its 830 us per chunk at config 4 were those stores, the loads + statistics 260,
//   the digit arithmetic 150.  The loads pay for it with 64-byte pieces (the four waves of the workgroup cover 256 contiguous bytes).
//   STATS: the digits are cut under the scales PREDICTED from the previous chunk of the same call (scale_in = pred[c], a power of two;
//   anything else: the column is not cut) while maxima / sums of |v| over the 256-row block go to part[block][2][kp] as
//   gen_colstat_kernel writes them (the sum in a fixed tree: per lane over its rows in order, the two tiles, then the row groups);
//   gen_colscale_kernel then accepts or rejects the prediction per column.  Saves the second read of wP (16.6 KB per sample at k = 64).
//   !STATS: the first chunk of a call -- its scales are known (scale_in = scale[c]), the chunk's guard (flags[1]) is honoured.
// Workgroup = 256 rows (two tiles) x 32 columns; grid (ceil(kp / 32), ceil(n / 256)).
// (stats is a run-time argument, the statistics are always computed: as a template parameter the instantiation without them came out
//  at 1 260 us per chunk against 570 with them -- its loads sank into the branch that cuts the digits)
__global__ __launch_bounds__(256) void gen_wdigits_lines_kernel(const double *V, int64_t n, int64_t ncpad, int kp, int64_t npad,
                                                                const double *scale_in, signed char *BtW, double *part, int *flags,
                                                                int stats) {
    if (stats) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            flags[1] = 0;
            flags[2] = 0;
        }
    } else {
        if (flags[1]) return;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int rg = lane >> 3, cc = lane & 7;
    const int c = blockIdx.x * 32 + 8 * wave + cc;
    const bool okc = c < kp;
    int shift = 0;
    bool cut = false;
    if (okc) {
        const double ps = scale_in[c];
        if (ps > 0.0 && ps < 1.0e300) {
            int ex;
            (void)frexp(ps, &ex);  // scale = 2^(ex - 1)
            shift = -(ex - 1);
            cut = true;
        }
    }
    double mx = 0.0, sm = 0.0;
    const int64_t ib = (int64_t)blockIdx.y * 256 + 16 * rg;
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (ib + e < n && okc) ? V[(ib + e) * kp + c] : 0.0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int64_t i0 = ib + 128 * it;
        double vn[16];
        if (it == 0) {  // (the second tile's rows are requested before the first tile's digits are cut)
#pragma unroll
            for (int e = 0; e < 16; ++e) vn[e] = (i0 + 128 + e < n && okc) ? V[(i0 + 128 + e) * kp + c] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double a = fabs(v[e]);
            mx = fmax(mx, a);  // (a NaN leaves the maximum alone; the sum carries it into the guard)
            sm += a;
        }
        if (cut && i0 < ncpad) {
            // I = rint(v 2^shift) as hi 2^28 + lo in 32-bit halves (the 64-bit form -- llrint, eight 64-bit add / shift pairs -- was 150 of
            // the kernel's 570 us): hi = rint(y 2^-28), lo = rint(y - hi 2^28) -- the difference is exact (a multiple of ulp(y) below
            // 2^27) and hi 2^28 is even, so hi 2^28 + lo = rint(y) including the ties.  Four balanced 7-bit digits from lo, its carry
            // into hi, four from hi (the last one takes what is left, as before).
            static_assert(GQS == 8, "two 28-bit halves");
            gi4_t q[GQS];
#pragma unroll
            for (int sl = 0; sl < GQS; ++sl) q[sl] = gi4_t{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const double y = ldexp(v[e], shift);
                const double hf = __builtin_rint(ldexp(y, -28));
                const double lf = __builtin_rint(y - ldexp(hf, 28));
                int lo = (int)lf, hi = (int)hf;
#pragma unroll
                for (int sl = 0; sl < GQS; ++sl) {
                    if (sl == 4) hi += lo;  // (what the four low digits left: -1, 0 or 1)
                    int &I = sl < 4 ? lo : hi;
                    const int t7 = I + 64;
                    const int dig = (t7 & 127) - 64;
                    I = t7 >> 7;
                    q[sl][e >> 2] |= (dig & 255) << (8 * (e & 3));
                }
            }
#pragma unroll
            for (int sl = 0; sl < GQS; ++sl) *reinterpret_cast<gi4_t *>(BtW + ((int64_t)sl * kp + c) * npad + i0) = q[sl];
        }
        if (it == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = vn[e];
        }
    }
    {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            mx = fmax(mx, __shfl_xor(mx, o));
            sm += __shfl_xor(sm, o);
        }
        if (stats && rg == 0 && okc) {
            part[((int64_t)blockIdx.y * 2) * kp + c] = mx;
            part[((int64_t)blockIdx.y * 2 + 1) * kp + c] = sm;
        }
    }
}

// C[M x N] (+)= scale[c] * sum_s 128^s ( A[M x K] . Bt[s][N x K]^T ), bytes, K a multiple of 64.
// Workgroup tile 128 rows x 32 columns x 8 slices; wave w: rows 32 w .. +31 (two 16-row tiles) x two 16-column
// tiles x 8 slices = 32 accumulator tiles.  Operands of K-step k+1 are fetched into registers before the MFMAs of
// step k and go to the other LDS buffer after them.
struct I8GemmArgs {
    const unsigned char *A;
    int64_t lda;
    const signed char *Bt;
    int64_t ldb;      // bytes per column row of one plane
    int64_t plane;    // bytes between slices
    int64_t M, N, K;
    const double *scale;
    double *out;
    int64_t ldo;
    int accumulate;
    const int *guard;  // run only if *guard == 0
    // split-K (blockIdx.z = slice z of nsplit): slice z covers [z ksplit, (z + 1) ksplit); slice 0 goes to `out`, slice
    // z >= 1 to the dense M x N buffer out2 + (z - 1) M N (overwritten), which add_partial_kernel then adds to `out`
    // in slice order -- a fixed order.  Used where the tile grid is a few workgroups more than the chip holds at once
    // (S at config 4: 520 tiles on 512 slots, two slices) and where it is far too small to fill it (S at d = 256,
    // k = 11: 6 tiles -- one step outside the fused kernel the statistics contraction ran on 12 of 256 CUs).
    int64_t ksplit;    // 0 = no split
    int nsplit;
    double *out2;
    // (round 5) XCD-aware tile order: a 1-D grid whose workgroup L (dispatched to XCD L % 8) takes column block (L % 8) + 8 j -- so
    // a column stripe of B is only ever read through ONE XCD's L2 -- and, within the XCD, the tiles of one row block side by
    // side (j fastest), then the next row block, then the next K-slice.  ncb / nrb: column / row blocks of the product.
    int xcd_map;
    int ncb, nrb;
    int tile_rows;     // 0: the launcher's choice; 128: the 128-row tile (two workgroups per CU)
};

// KB = bytes of K per staged step (64 or 128: whole 128-byte lines per row at 128); rows padded by 16 B in LDS.
// TM = rows of the workgroup tile: 128 (4 waves, two workgroups per CU) or 256 (8 waves, one workgroup per CU: the
// B stripes -- the 8 digit planes -- are re-read by half as many row blocks).
// BUF: operands through buffer descriptors (per-thread piece offsets computed once, the K position a scalar offset:
// no vector address arithmetic in the K loop -- the pointer form spent 2.7 vector instructions per MFMA on 64-bit
// addresses and bounds, SQ_INSTS_VALU 338 M against SQ_INSTS_MFMA 91 M per launch); needs both operands < 2 GiB.
// The K loop of the buffer form runs WITHOUT a workgroup barrier (round 5, late; I8_RING=0: the two-buffer loop of rounds 2-5 with its
// barrier per step).  The LDS image is a ring of K-steps whose stages carry two monotonic LDS counters: `full` (waves that have stored
// their pieces of the step) and `empty` (waves that have requested their fragments of it).  A wave signals a stage full at least one whole
// step before anyone waits for it and signals it empty as soon as its fragment reads are issued (a wave's LDS operations execute in
// order); the counters of the next step are read in the middle of a step's MFMAs, so the usual case costs no LDS round trip.
//   256-row tile (one workgroup per CU): four stages, a step is stored into the stage freed TWO steps ago (a fast wave may run two
//     steps ahead of the slowest), and the operands are requested as WHOLE 128-byte lines, two K-steps at a time (I8_WIDE: a K-step
//     takes 64 bytes of each of 512 rows that lie far apart -- 64 KB of half-used lines against 32 KB of L1 -- so the second halves
//     were fetched from L2 again; eight lanes now cover a row's 128 bytes, and the lane -> piece map alternates between a thread's row
//     blocks so that every thread holds pieces of both steps and each step's staging is a full-width ds_write).
//   128-row tile (two workgroups per CU): three stages, one step of slack, 64-byte requests (the wide form spills there).
// Measured at config 4 (profiles/r05/i8gemm_ring_ab.log, i8gemm_pmc.log): 174.2 -> 168.9 ms per EM iteration; per launch the Gram
// product 1 780 -> 1 680 us, the statistics product 1 275 -> 1 190 us.  What did NOT pay, each built and parity-green
// (tools/experiments/i8gemm_ring_variants.patch): a three-stage ring at one step of slack (178 ms -- slower than the barrier), a
// half-step stagger of the second wave of every SIMD (+-0), A fragments straight from global memory into the operand registers
// (195 ms), the mask operand as packed BITS expanded by the VALU (187 ms: A costs no LDS and an eighth of the bytes, and the 48
// VALU operations per step cost more than both).
#ifndef I8_RING
#define I8_RING 1
#endif
#ifndef I8_WIDE
#define I8_WIDE 1
#endif
#ifndef I8_STAGES
#define I8_STAGES 1  // measured at config 4 (round 2): 1 -> 254.0, 2 -> 253.4, 3 -> 253.2, 4 -> 262.4 ms (spills); the registers go to the fragments
#endif
template <int KB, int TM = 128, bool BUF = true>
__global__ __launch_bounds__(2 * TM, 2) void i8gemm_kernel(I8GemmArgs g) {  // two waves per SIMD (<= 256 registers): measured 64 vs 76 ms per iteration at 1 wave, 118 ms at 3 (spills)
    constexpr int THREADS = 2 * TM;
    constexpr int PR = KB / 16;                         // 16-byte pieces per row of a K-step
    constexpr int NA = TM * PR / THREADS, NB = GQS * 32 * PR / THREADS;  // pieces per thread
    static_assert(KB == 64, "the LDS swizzle below is written for four pieces per row");
    // LDS image of a K-step: PIECE-major, [4 pieces][rows][16 B], the row's low four bits XORed with g(piece) =
    // {0, 2, 12, 14}.  ds_read_b128 serves a wave in four groups of 16 lanes -- rows {0-3, 12-15} of piece p with rows
    // 4-11 of piece p + 1 -- and ds_write_b128 in groups of 8 lanes (two rows x four pieces here): with this image
    // both hit 16 (8) different 16-byte slots.  The row-major image with 80-byte rows it replaces had two-way
    // conflicts on the reads (5 r + p collides between the two halves of a group): SQ_LDS_BANK_CONFLICT was
    // 0.9-1.1e9 against SQ_ACTIVE_INST_LDS 0.4-0.6e9 per launch (round 2, gpurun_out/pmcgen).
    extern __shared__ __attribute__((aligned(16))) unsigned char i8sm[];
    constexpr int AROWS = TM, BROWS = GQS * 32;
    constexpr int NBUF = (BUF && I8_RING != 0) ? (TM == 256 ? 4 : 3) : 2;  // stages of the LDS image
    constexpr int LAG = TM == 256 ? 2 : 1;                                  // a step is stored into the stage freed LAG steps ago
    constexpr bool WIDE = BUF && I8_RING != 0 && I8_WIDE != 0 && TM == 256;
    unsigned char *As = i8sm;                              // [NBUF][4][TM][16]
    unsigned char *Bs = i8sm + NBUF * 4 * AROWS * 16;  // [NBUF][4][GQS * 32][16]
    auto slot = [](int rows, int buf, int row, int q) {    // byte offset of (row, piece q) in buffer buf
        const int g = (q & 1) * 2 + (q >> 1) * 12;         // 0, 2, 12, 14
        return ((buf * 4 + q) * rows + (row ^ g)) * 16;
    };
    if (g.guard && *g.guard != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // (an XCD-aware order of the tiles -- column block c pinned to XCD c % 8, row blocks of one column block side by
    //  side -- measured SLOWER, 106 vs 97 ms per iteration at N = 400 k, d = 1024, k = 64: plain 2-D order kept)
    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    if (g.xcd_map) {
        const int L = (int)blockIdx.x, xcd = L & 7, sq = L >> 3;
        const int ncmax = (g.ncb + 7) >> 3, per_z = ncmax * g.nrb;
        bz = sq / per_z;
        const int s2 = sq - bz * per_z;
        by = s2 / ncmax;
        bx = xcd + 8 * (s2 - by * ncmax);
        if (bx >= g.ncb) return;  // (the XCDs with one column block fewer)
    }
    const int64_t m0 = (int64_t)by * TM;
    const int64_t n0 = (int64_t)bx * 32;
    const int zs = g.ksplit > 0 ? bz : 0;
    const bool second = zs > 0;
    const int64_t kbeg = (int64_t)zs * g.ksplit;
    const int64_t kend = g.ksplit > 0 ? (kbeg + g.ksplit < g.K ? kbeg + g.ksplit : g.K) : g.K;
    // Wave tile 64 rows x 16 columns x GQS slices (waves as 2 x 2 over the 128 x 32 workgroup tile): 4 A fragments
    // + GQS B fragments read from LDS per 4 GQS MFMAs of a 64-byte K-step -- 0.375 reads per MFMA where a 32 x 32
    // wave tile needs 0.56 (measured: the same 266 ms per iteration at config 4 either way -- LDS reads are not what
    // holds this kernel at a quarter of the int8 MFMA rate).
    const int wm = wave >> 1, wn = wave & 1;
    gi4_t acc[4][GQS];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int s = 0; s < GQS; ++s) acc[a][s] = gi4_t{0, 0, 0, 0};
    constexpr int ST = BUF ? I8_STAGES : 1;  // K-steps in flight in registers (buffer form: over-fetching past K is harmless)
    gi4_t ra[ST][NA], rb[ST][NB];
    // rows / columns outside the product read as zeros: their offset lies past the descriptor's extent
    unsigned aoff[NA], boff[NB];
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(g.A), 0, BUF ? (int)(g.M * g.lda) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<signed char *>(g.Bt), 0, BUF ? (int)(GQS * g.plane) : 0, 0x00020000);
    if constexpr (BUF) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int piece = tid + THREADS * u, r = piece / PR, q = piece % PR;
            const int64_t row = m0 + r;
            aoff[u] = row < g.M ? (unsigned)(row * g.lda + 16 * q) : 0x80000000u;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int piece = tid + THREADS * u, q = piece % PR, rr = piece / PR, cc = rr & 31, sl = rr >> 5;
            const int64_t col = n0 + cc;
            boff[u] = col < g.N ? (unsigned)(sl * g.plane + col * g.ldb + 16 * q) : 0x80000000u;
        }
    }
    auto fetch = [&](int64_t k0, int st) {
        if constexpr (BUF) {  // K is a whole number of KB-byte steps (rows padded to 64 bytes)
            typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)aoff[u], (int)k0, 0);
                ra[st][u] = gi4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)boff[u], (int)k0, 0);
                rb[st][u] = gi4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int piece = tid + THREADS * u, r = piece / PR, q = piece % PR;
            const int64_t row = m0 + r;
            ra[st][u] = (row < g.M && k0 + 16 * q < kend) ? *reinterpret_cast<const gi4_t *>(g.A + row * g.lda + k0 + 16 * q)
                                                     : gi4_t{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int piece = tid + THREADS * u, q = piece % PR, rr = piece / PR, cc = rr & 31, sl = rr >> 5;
            const int64_t col = n0 + cc;
            rb[st][u] = (col < g.N && k0 + 16 * q < kend)
                        ? *reinterpret_cast<const gi4_t *>(g.Bt + sl * g.plane + col * g.ldb + k0 + 16 * q)
                        : gi4_t{0, 0, 0, 0};
        }
    };
    auto stash = [&](int buf, int st) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int piece = tid + THREADS * u;
            *reinterpret_cast<gi4_t *>(As + slot(AROWS, buf, piece / PR, piece % PR)) = ra[st][u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int piece = tid + THREADS * u;
            *reinterpret_cast<gi4_t *>(Bs + slot(BROWS, buf, piece / PR, piece % PR)) = rb[st][u];
        }
    };
    auto compute = [&](int buf) {
        // every fragment of the K-step is requested before the first MFMA (12 reads in flight: the LDS latency is paid
        // once per step, not once per slice pair as hipcc schedules the interleaved form)
        gi4_t fa[4], fb[GQS];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const gi4_t *>(As + slot(AROWS, buf, 64 * wm + 16 * a + l15, l4));
#pragma unroll
        for (int s = 0; s < GQS; ++s) fb[s] = *reinterpret_cast<const gi4_t *>(Bs + slot(BROWS, buf, s * 32 + 16 * wn + l15, l4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < GQS; ++s)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[s], acc[a][s], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (BUF) {
        // ST K-steps travel in registers while one is contracted out of LDS: a step is requested ST iterations (ST x
        // ~1 k cycles of MFMAs per SIMD) before it is staged.  (The MFMA pipe is 30 % busy in this kernel by
        // SQ_VALU_MFMA_BUSY_CYCLES; a deeper pipeline did not change that: memory latency is not what it waits for.)
        const int nsteps = (int)((kend - kbeg) / KB);
        if constexpr (NBUF >= 3) {
            constexpr unsigned NW = THREADS / 64;
            unsigned *full = reinterpret_cast<unsigned *>(i8sm + NBUF * 4 * (AROWS + BROWS) * 16), *empty = full + NBUF;
            if (tid < 2 * NBUF) full[tid] = 0u;
            __syncthreads();
            auto signal = [&](unsigned *ctr) {
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
            };
            auto await = [&](const unsigned *ctr, unsigned need) {
                for (;;) {
                    const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if ((int)(seen - need) >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            };
            if constexpr (WIDE) {
                static_assert(!WIDE || (NBUF == 4 && LAG == 2 && ST == 1), "the wide request walks pairs of steps over a four-stage ring");
                constexpr int RPT = THREADS / 8, NA2 = TM * 8 / THREADS, NB2 = BROWS * 8 / THREADS;
                static_assert(NA2 % 2 == 0 && NB2 % 2 == 0 && RPT % 16 == 0, "row blocks come in pairs");
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const int tr = tid >> 3, t8 = tid & 7;
                const bool selhi = (t8 & 4) != 0;  // this thread's even row blocks hold the second step of a pair
                gi4_t wa[NA2], wb[NB2];
                unsigned woa[NA2], wob[NB2];
#pragma unroll
                for (int u = 0; u < NA2; ++u) {
                    const int r = RPT * u + ((u & 1) ? ((tr + 1) & (RPT - 1)) : tr), q8 = t8 ^ ((u & 1) << 2);
                    const int64_t row = m0 + r;
                    woa[u] = row < g.M ? (unsigned)(row * g.lda + 16 * q8) : 0x80000000u;
                }
#pragma unroll
                for (int u = 0; u < NB2; ++u) {
                    const int r = RPT * u + ((u & 1) ? ((tr + 1) & (RPT - 1)) : tr), q8 = t8 ^ ((u & 1) << 2);
                    const int64_t col = n0 + (r & 31);
                    wob[u] = col < g.N ? (unsigned)((r >> 5) * g.plane + col * g.ldb + 16 * q8) : 0x80000000u;
                }
                // where this lane's piece of half h goes inside a stage: row block 2 j + (selhi ^ h), piece t8 & 3
                const int qq = t8 & 3, gq = (qq & 1) * 2 + (qq >> 1) * 12;
                int la[2], lb[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool odd = selhi != (h != 0);
                    const int rb = odd ? RPT + ((tr + 1) & (RPT - 1)) : tr;
                    la[h] = (qq * AROWS + (rb ^ gq)) * 16;
                    lb[h] = (qq * BROWS + (rb ^ gq)) * 16;
                }
                auto fetchw = [&](int64_t k0) {
#pragma unroll
                    for (int u = 0; u < NA2; ++u) {
                        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(ars, (int)woa[u], (int)k0, 0);
                        wa[u] = gi4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
                    }
#pragma unroll
                    for (int u = 0; u < NB2; ++u) {
                        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(brs, (int)wob[u], (int)k0, 0);
                        wb[u] = gi4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
                    }
                };
                auto stashw = [&](int h, int stage) {
                    const bool odd = selhi != (h != 0);
                    unsigned char *ad = As + stage * (4 * AROWS * 16) + la[h], *bd = Bs + stage * (4 * BROWS * 16) + lb[h];
#pragma unroll
                    for (int j = 0; j < NA2 / 2; ++j) {
                        gi4_t v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = odd ? wa[2 * j + 1][e] : wa[2 * j][e];
                        *reinterpret_cast<gi4_t *>(ad + 2 * RPT * j * 16) = v;
                    }
#pragma unroll
                    for (int j = 0; j < NB2 / 2; ++j) {
                        gi4_t v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = odd ? wb[2 * j + 1][e] : wb[2 * j][e];
                        *reinterpret_cast<gi4_t *>(bd + 2 * RPT * j * 16) = v;
                    }
                };
                fetchw(kbeg);
                stashw(0, 0);
                signal(full + 0);
                stashw(1, 1);
                signal(full + 1);
                fetchw(kbeg + 2 * KB);
                unsigned pf = 0u, pe = 0u;
                auto step = [&](int i, int h) {  // h = i & 1: step q = i + 2 is half h of its pair
                    const int st = i & 3;
                    const unsigned nf = NW * (unsigned)((i >> 2) + 1);
                    if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)pf) - nf) < 0) await(full + st, nf);
                    gi4_t fa[4], fb[GQS];
#pragma unroll
                    for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const gi4_t *>(As + slot(AROWS, st, 64 * wm + 16 * a + l15, l4));
#pragma unroll
                    for (int s = 0; s < GQS; ++s) fb[s] = *reinterpret_cast<const gi4_t *>(Bs + slot(BROWS, st, s * 32 + 16 * wn + l15, l4));
                    signal(empty + st);
                    __builtin_amdgcn_sched_barrier(0);
                    const int q = i + 2, sq = q & 3;
                    if (q < nsteps) {  // (uniform)
                        const unsigned ne = NW * (unsigned)(q >> 2);
                        if (i >= 2 && (int)((unsigned)__builtin_amdgcn_readfirstlane((int)pe) - ne) < 0) await(empty + sq, ne);
                        stashw(h, sq);
                        signal(full + sq);
                        if (h == 1) fetchw(kbeg + (int64_t)(q + 1) * KB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < GQS / 2; ++s)
#pragma unroll
                        for (int a = 0; a < 4; ++a) acc[a][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[s], acc[a][s], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    pf = __hip_atomic_load(full + ((st + 1) & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pe = __hip_atomic_load(empty + ((sq + 1) & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = GQS / 2; s < GQS; ++s)
#pragma unroll
                        for (int a = 0; a < 4; ++a) acc[a][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[s], acc[a][s], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                for (int i = 0; i < nsteps; i += 2) {
                    step(i, 0);
                    if (i + 1 < nsteps) step(i + 1, 1);  // (uniform)
                }
            } else {
            // LAG: the stage a step is stored into was freed LAG steps ago (a fast wave may run LAG steps ahead of the slowest one);
            // steps 0 .. NBUF - LAG - 1 staged up front, steps NBUF - LAG .. NBUF - LAG - 1 + ST in registers.
            static_assert(LAG >= 1 && LAG < NBUF, "ring lag");
#pragma unroll
            for (int j = 0; j < NBUF - LAG; ++j) {
                fetch(kbeg + (int64_t)j * KB, 0);
                stash(j, 0);
                signal(full + j);
            }
#pragma unroll
            for (int j = 0; j < ST; ++j) fetch(kbeg + (int64_t)(NBUF - LAG + j) * KB, j);
            // the counters of the NEXT step are read in the middle of a step's MFMAs and looked at when that step begins: the usual case
            // (everyone has arrived) costs no LDS round trip at the head of the step
            unsigned pf = 0u, pe = 0u;
            auto step = [&](int i, int rs) {
                const int st = i % NBUF;
                const unsigned nf = NW * (unsigned)(i / NBUF + 1);
                if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)pf) - nf) < 0) await(full + st, nf);
                gi4_t fa[4], fb[GQS];
#pragma unroll
                for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const gi4_t *>(As + slot(AROWS, st, 64 * wm + 16 * a + l15, l4));
#pragma unroll
                for (int s = 0; s < GQS; ++s) fb[s] = *reinterpret_cast<const gi4_t *>(Bs + slot(BROWS, st, s * 32 + 16 * wn + l15, l4));
                signal(empty + st);        // (the reads above are ahead of this add in the wave's LDS queue)
                __builtin_amdgcn_sched_barrier(0);
                // step q (register set rs) goes to the stage step i - LAG was read from, behind the fragment requests
                const int q = i + NBUF - LAG, sq = q % NBUF;
                if (q < nsteps) {  // (uniform)
                    const unsigned ne = NW * (unsigned)(q / NBUF);
                    if (i >= LAG && (int)((unsigned)__builtin_amdgcn_readfirstlane((int)pe) - ne) < 0) await(empty + sq, ne);
                    stash(sq, rs);
                    signal(full + sq);
                    fetch(kbeg + (int64_t)(q + ST) * KB, rs);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < GQS / 2; ++s)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[s], acc[a][s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                pf = __hip_atomic_load(full + (st + 1 == NBUF ? 0 : st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pe = __hip_atomic_load(empty + (sq + 1 == NBUF ? 0 : sq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = GQS / 2; s < GQS; ++s)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[s], acc[a][s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int i = 0; i < nsteps; i += ST) {
#pragma unroll
                for (int j = 0; j < ST; ++j)
                    if (i + j < nsteps) step(i + j, j);  // (uniform)
            }
            }
        } else {
#pragma unroll
        for (int j = 0; j < ST; ++j) fetch(kbeg + (int64_t)j * KB, j);
        stash(0, 0);
        fetch(kbeg + (int64_t)ST * KB, 0);
        __syncthreads();
        for (int i = 0; i < nsteps; i += ST) {
#pragma unroll
            for (int j = 0; j < ST; ++j) {
                if (i + j < nsteps) {  // (uniform)
                    const int buf = (i + j) & 1;
                    compute(buf);
                    stash(buf ^ 1, (j + 1) % ST);                                       // step i + j + 1
                    fetch(kbeg + (int64_t)(i + j + 1 + ST) * KB, (j + 1) % ST);        // step i + j + 1 + ST
                    __syncthreads();
                }
            }
        }
        }
    } else {
        fetch(kbeg, 0);
        stash(0, 0);
        __syncthreads();
        int buf = 0;
        for (int64_t k0 = kbeg; k0 < kend; k0 += KB) {
            const bool more = k0 + KB < kend;
            if (more) fetch(k0 + KB, 0);
            compute(buf);
            if (more) stash(buf ^ 1, 0);
            __syncthreads();
            buf ^= 1;
        }
    }
    {
        const int64_t col = n0 + 16 * wn + l15;
        if (col >= g.N) return;
        const double sc = g.scale[col];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                const int64_t row = m0 + 64 * wm + 16 * a + 4 * l4 + r;
                if (row >= g.M) continue;
                double v = (double)acc[a][GQS - 1][r];
#pragma unroll
                for (int s = GQS - 2; s >= 0; --s) v = v * 128.0 + (double)acc[a][s][r];
                if (second) {
                    g.out2[(int64_t)(zs - 1) * g.M * g.N + row * g.N + col] = v * sc;
                } else {
                    double *dst = g.out + row * g.ldo + col;
                    *dst = g.accumulate ? *dst + v * sc : v * sc;
                }
            }
    }
}

// ------------------------------------------------------------------ wave-level SPD tools (runtime k <= 64)
// Matrix in LDS, row-major with leading dimension LD; lane r owns row r.  Plain (non-volatile) LDS
// pointers so the inner loops can be unrolled and their loads batched; lanes exchange data through LDS
// only at the explicit wave_sync() points (LDS operations of one wave execute in issue order; the fence
// stops hipcc from moving accesses across the point).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The factor overwrites the lower triangle, diagonal slots hold 1/L_pp.  Returns false when a pivot is
// not positive.
__device__ bool wave_cholesky(double *Mw, int k, int LD, int lane, double &logdet) {
    bool ok = true;
    double mant = 1.0;
    int ex = 0;
    wave_sync();
    for (int p = 0; p < k; ++p) {
        const double piv = Mw[p * LD + p];
        ok = ok && (piv > 0.0) && (piv < 1.0e308);
        const double rinv = 1.0 / sqrt(piv);
        int e;
        mant *= frexp(piv, &e);
        ex += e;
        double lrp = 0.0;
        wave_sync();  // everyone has read the pivot before its slot is overwritten
        if (lane > p && lane < k) {
            lrp = Mw[lane * LD + p] * rinv;
            Mw[lane * LD + p] = lrp;
        }
        if (lane == p) Mw[p * LD + p] = rinv;
        wave_sync();  // column p is final
        double *myrow = Mw + (lane < k ? lane : 0) * LD;
#pragma unroll 8
        for (int c = p + 1; c < k; ++c) {
            const double lcp = Mw[c * LD + p];
            if (lane >= c && lane < k) myrow[c] -= lrp * lcp;
        }
        wave_sync();  // trailing update visible before the next pivot is read
    }
    logdet = log(mant) + (double)ex * LN_2;
    return ok;
}

// Solve (L L^T) z = v, lane r holds v_r on entry and z_r on exit; quad = |L^-1 v|^2.
__device__ double wave_chol_solve(const double *Mw, int k, int LD, int lane, double v, double &quad) {
    const int lr = lane < k ? lane : 0;
    for (int p = 0; p < k; ++p) {
        const double yp = __shfl(v, p, 64) * Mw[p * LD + p];
        if (lane == p) v = yp;
        if (lane > p && lane < k) v -= Mw[lr * LD + p] * yp;
    }
    quad = gwave_sum(lane < k ? v * v : 0.0);
    for (int p = k - 1; p >= 0; --p) {
        const double zp = __shfl(v, p, 64) * Mw[p * LD + p];
        if (lane == p) v = zp;
        if (lane < p) v -= Mw[p * LD + lane] * zp;
    }
    return v;
}

// Uw[a][c] = (M^-1)_{ac}; lane c owns column c (two triangular solves per column, lanes independent:
// Mw is read-only here and each lane touches only its own column of Uw, so no wave_sync is needed).
__device__ void wave_chol_inverse(const double *Mw, double *Uw, int k, int LD, int lane) {
    const int c = lane < k ? lane : k - 1;
    double *ucol = Uw + c;
    for (int a = 0; a < k; ++a) {
        double s0 = (a == c) ? 1.0 : 0.0, s1 = 0.0;
        const double *mrow = Mw + a * LD;
        int t = 0;
#pragma unroll 4
        for (; t + 2 <= a; t += 2) {  // two partial sums: shorter dependency chains
            s0 -= mrow[t] * ucol[t * LD];
            s1 -= mrow[t + 1] * ucol[(t + 1) * LD];
        }
        if (t < a) s0 -= mrow[t] * ucol[t * LD];
        if (lane < k) ucol[a * LD] = (a >= c) ? (s0 + s1) * mrow[a] : 0.0;
    }
    for (int a = k - 1; a >= 0; --a) {
        double s0 = ucol[a * LD], s1 = 0.0;
        int t = a + 1;
#pragma unroll 4
        for (; t + 2 <= k; t += 2) {
            s0 -= Mw[t * LD + a] * ucol[t * LD];
            s1 -= Mw[(t + 1) * LD + a] * ucol[(t + 1) * LD];
        }
        if (t < k) s0 -= Mw[t * LD + a] * ucol[t * LD];
        if (lane < k) ucol[a * LD] = (s0 + s1) * Mw[a * LD + a];
    }
}

// ------------------------------------------------------------------ per-sample solve
// (SolveArgs: ppca_solve.hpp)

__global__ __launch_bounds__(128) void solve_kernel(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = a.k, kp = k * (k + 1) / 2, LD = k | 1;
    double *Mw = gsm + (size_t)wave * (2 * k * LD + 64);
    double *Uw = Mw + k * LD;
    double *zw = Uw + k * LD;
    const double s2 = a.model[1], lnsig = a.model[2];
    const int64_t stride = (int64_t)gridDim.x * 2;
    for (int64_t i = (int64_t)blockIdx.x * 2 + wave; i < a.n; i += stride) {
        double *g = a.G + i * kp;
        double *bz = a.Bz + i * (k + 1);
        wave_sync();  // previous sample's reads of Mw/Uw/zw are done
        for (int e = lane; e < kp; e += 64) {
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            int c = e - r * (r + 1) / 2;
            Mw[r * LD + c] = g[e] + (r == c ? s2 : 0.0);
        }
        const double bv = lane < k ? bz[lane] : 0.0;
        double logdet, quad;
        wave_cholesky(Mw, k, LD, lane, logdet);
        const double z = wave_chol_solve(Mw, k, LD, lane, bv, quad);
        if (lane < k) zw[lane] = z;
        const double zz = gwave_sum(lane < k ? z * z : 0.0);
        wave_chol_inverse(Mw, Uw, k, LD, lane);
        wave_sync();  // zw and Uw columns are read across lanes below
        const double tr = gwave_sum(lane < k ? Uw[lane * LD + lane] : 0.0);
        const double wgt = a.w ? a.w[i] : 1.0;
        const double xx = a.xx[i];
        const int m = (int)a.mc[i];
        const double lk = sample_llk(xx, quad, logdet, s2, lnsig, m, k);
        if (a.em) {
            // w P = w (z z^T + s2 M^-1), packed; lane c writes column c of every row a >= c
            if (lane < k) {
                for (int r = lane; r < k; ++r) g[r * (r + 1) / 2 + lane] = wgt * (zw[r] * z + s2 * Uw[r * LD + lane]);
                bz[lane] = wgt * z;
            }
            if (lane == 0) {
                bz[k] = wgt;
                double *sc = a.sc + i * 4;
                sc[0] = m > 0 ? wgt * s2 * ((double)k - s2 * tr) : 0.0;
                sc[1] = m > 0 ? wgt * (xx - quad - s2 * zz) : 0.0;
                sc[2] = wgt * lk;
                sc[3] = m > 0 ? 1.0 : 0.0;
            }
        } else {
            if (lane < k) {
                bz[lane] = z;  // unweighted state for the reconstruction pass
                if (a.states) a.states[i * k + lane] = z;
                for (int r = 0; r < k; ++r) {
                    const double sv = s2 * Uw[r * LD + lane];
                    if (a.covs) a.covs[(i * k + r) * k + lane] = sv;
                    if (r >= lane) g[r * (r + 1) / 2 + lane] = sv;  // Sigma packed, for covariance diagonals
                }
            }
            if (lane == 0) {
                double *sc = a.sc + i * 4;
                sc[0] = 0.0;
                sc[1] = 0.0;
                sc[2] = wgt * lk;
                sc[3] = 0.0;
                if (a.llks) a.llks[i] = lk;
            }
        }
    }
}

// ------------------------------------------------------------------ state sizes 65 .. GENERIC_MAX_K (round 5): one WORKGROUP per matrix
// The reference bounds the state size nowhere (ppca_model.rs:51-70, output_covariance.rs:57-70); rounds 1-4 refused k > 64 (a
// wave's 64 lanes owned the rows of M).  A correct path without a performance claim: 256 threads, thread r < k owns row r of
// M in LDS (k x (k | 1) doubles: 132 KB at k = 128), __syncthreads() where the wave forms have wave_sync().  The Cholesky
// factor is inverted IN PLACE (row r of T = L^-1 from rows < r of T and row r of L), and M^-1 = T^T T is formed entry by entry
// on the way out, so one matrix is all the LDS the solve needs.  Same I/O contract as solve_kernel.
__device__ bool blk_cholesky(double *M, int k, int LD, int tid, double &logdet) {
    bool ok = true;
    double mant = 1.0;
    int ex = 0;
    __syncthreads();
    for (int p = 0; p < k; ++p) {
        const double piv = M[p * LD + p];
        ok = ok && (piv > 0.0) && (piv < 1.0e308);
        const double rinv = 1.0 / sqrt(piv);
        int e;
        mant *= frexp(piv, &e);
        ex += e;
        double lrp = 0.0;
        __syncthreads();  // everyone has read the pivot before its slot is overwritten
        if (tid > p && tid < k) {
            lrp = M[tid * LD + p] * rinv;
            M[tid * LD + p] = lrp;
        }
        if (tid == p) M[p * LD + p] = rinv;
        __syncthreads();  // column p is final
        if (tid > p && tid < k) {
            double *myrow = M + tid * LD;
            for (int c = p + 1; c <= tid; ++c) myrow[c] -= lrp * M[c * LD + p];
        }
        __syncthreads();  // trailing update visible before the next pivot is read
    }
    logdet = log(mant) + (double)ex * LN_2;
    return ok;
}
// (L L^T) z = v with v, z in the LDS vector vs (thread r owns entry r); quad = |L^-1 v|^2 (red: 256 doubles of scratch)
__device__ double blk_chol_solve(const double *M, int k, int LD, int tid, double *vs, double *red) {
    for (int p = 0; p < k; ++p) {
        __syncthreads();
        const double yp = vs[p] * M[p * LD + p];
        __syncthreads();
        if (tid == p) vs[p] = yp;
        if (tid > p && tid < k) vs[tid] -= M[tid * LD + p] * yp;
    }
    __syncthreads();
    red[tid] = tid < k ? vs[tid] * vs[tid] : 0.0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double quad = red[0];
    for (int p = k - 1; p >= 0; --p) {
        __syncthreads();
        const double zp = vs[p] * M[p * LD + p];
        __syncthreads();
        if (tid == p) vs[p] = zp;
        if (tid < p) vs[tid] -= M[p * LD + tid] * zp;
    }
    __syncthreads();
    return quad;
}
// L (strict lower triangle, 1 / L_pp on the diagonal) -> T = L^-1 in place
__device__ void blk_tri_inverse(double *M, int k, int LD, int tid) {
    for (int r = 1; r < k; ++r) {
        double v = 0.0;
        if (tid < r) {
            double s0 = 0.0;
            for (int t = tid; t < r; ++t) s0 += M[r * LD + t] * (t == tid ? M[t * LD + t] : M[t * LD + tid]);
            v = -s0 * M[r * LD + r];
        }
        __syncthreads();  // everyone has read row r of L
        if (tid < r) M[r * LD + tid] = v;
        __syncthreads();
    }
}
__device__ __forceinline__ double blk_sum(double v, double *red, int tid) {
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    return red[0];
}
__global__ __launch_bounds__(256) void solve_big_kernel(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int tid = threadIdx.x;
    const int k = a.k, kp = k * (k + 1) / 2, LD = k | 1;
    double *M = gsm, *vs = M + (size_t)k * LD, *zs = vs + k, *red = zs + k;
    const double s2 = a.model[1], lnsig = a.model[2];
    for (int64_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        double *g = a.G + i * kp;
        double *bz = a.Bz + i * (k + 1);
        __syncthreads();  // the previous sample's reads of M / vs / zs are done
        for (int r = tid; r < k; r += 256)
            for (int c = 0; c <= r; ++c) M[r * LD + c] = g[r * (r + 1) / 2 + c] + (r == c ? s2 : 0.0);
        if (tid < k) vs[tid] = bz[tid];
        double logdet;
        blk_cholesky(M, k, LD, tid, logdet);
        const double quad = blk_chol_solve(M, k, LD, tid, vs, red);
        const double z = tid < k ? vs[tid] : 0.0;
        if (tid < k) zs[tid] = z;
        const double zz = blk_sum(z * z, red, tid);
        blk_tri_inverse(M, k, LD, tid);
        // (M^-1)_rc = sum_{t >= r} T_tr T_tc for r >= c; the diagonal first, for the trace
        double trp = 0.0;
        if (tid < k) {
            double s0 = 0.0;
            for (int t = tid; t < k; ++t) s0 += M[t * LD + tid] * M[t * LD + tid];
            trp = s0;
        }
        const double tr = blk_sum(trp, red, tid);
        const double wgt = a.w ? a.w[i] : 1.0;
        const double xx = a.xx[i];
        const int m = (int)a.mc[i];
        const double lk = sample_llk(xx, quad, logdet, s2, lnsig, m, k);
        for (int e = tid; e < kp; e += 256) {
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            const int c = e - r * (r + 1) / 2;
            double s0 = 0.0;
            for (int t = r; t < k; ++t) s0 += M[t * LD + r] * M[t * LD + c];
            if (a.em) {
                g[e] = wgt * (zs[r] * zs[c] + s2 * s0);  // w P = w (z z^T + s2 M^-1), packed
            } else {
                const double sv = s2 * s0;
                g[e] = sv;  // Sigma packed, for the covariance diagonals
                if (a.covs) {
                    a.covs[(i * k + r) * k + c] = sv;
                    a.covs[(i * k + c) * k + r] = sv;
                }
            }
        }
        if (tid < k) {
            bz[tid] = a.em ? wgt * z : z;
            if (!a.em && a.states) a.states[i * k + tid] = z;
        }
        if (tid == 0) {
            double *sc = a.sc + i * 4;
            if (a.em) {
                bz[k] = wgt;
                sc[0] = m > 0 ? wgt * s2 * ((double)k - s2 * tr) : 0.0;
                sc[1] = m > 0 ? wgt * (xx - quad - s2 * zz) : 0.0;
                sc[2] = wgt * lk;
                sc[3] = m > 0 ? 1.0 : 0.0;
            } else {
                sc[0] = 0.0;
                sc[1] = 0.0;
                sc[2] = wgt * lk;
                sc[3] = 0.0;
                if (a.llks) a.llks[i] = lk;
            }
        }
    }
}
static size_t solve_big_lds(int k) { return sizeof(double) * ((size_t)k * (k | 1) + 2 * (size_t)k + 256); }
static hipError_t launch_solve_big(const SolveArgs &a, int n_cu, hipStream_t s) {
    const size_t lds = solve_big_lds(a.k);
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        e != hipSuccess)
        return e;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(a.n, (int64_t)n_cu));
    hipLaunchKernelGGL(solve_big_kernel, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}
// finalisation's row systems at k > 64: one workgroup per dimension
__global__ __launch_bounds__(256) void gen_rowsolve_big_kernel(const double *stats, const double *min, double *mout, int d, int k, double tau) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int tid = threadIdx.x;
    const int kp = k * (k + 1) / 2, LD = k | 1;
    double *M = gsm, *vs = M + (size_t)k * LD, *red = vs + 2 * k;
    const StatsLayout L(d, k);
    for (int j = blockIdx.x; j < d; j += gridDim.x) {
        const double *S = stats + L.S + (int64_t)j * kp;
        __syncthreads();
        for (int r = tid; r < k; r += 256)
            for (int c = 0; c <= r; ++c) M[r * LD + c] = S[r * (r + 1) / 2 + c] + (r == c ? tau : 0.0);
        if (tid < k) vs[tid] = stats[L.cross + (int64_t)j * k + tid];
        double logdet;
        const bool ok = blk_cholesky(M, k, LD, tid, logdet);
        (void)blk_chol_solve(M, k, LD, tid, vs, red);
        if (tid < k) {
            const double old = min[MODEL_HDR + (int64_t)j * k + tid];
            mout[MODEL_HDR + (int64_t)j * k + tid] = ok ? vs[tid] : old;  // :313-321 keep the old row
        }
    }
}

// ------------------------------------------------------------------ per-sample solve, one LANE per sample (k <= 16)
// The blocked / broadcast solvers below give a whole wave to one sample: right at k = 64, a 70-fold waste at k = 11
// (13 k cycles per sample and wave: 6.4 ms of a 16 ms chunk one step outside the fused kernel).  Up to k = 16 the
// packed Cholesky factor (136 doubles) fits a lane's registers, so the fused kernel's per-sample code (Posterior<K>,
// ppca_small.hpp) runs here as it does there: 64 samples per wave instruction.  Same I/O contract as solve_kernel.
template <int K, bool EM>
__device__ __forceinline__ void solve_lane_body(const SolveArgs &a) {
    constexpr int KP = K * (K + 1) / 2;
    const double s2 = a.model[1], lnsig = a.model[2];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        double *g = a.G + i * KP;
        double *bz = a.Bz + i * (K + 1);
        Posterior<K> post;
        double pm;
        int pe;
        post.factor([&](int e) { return g[e]; }, s2, pm, pe);
        double z[K], quad, zz;
        post.solve([&](int c) { return bz[c]; }, z, quad, zz);
        const double wgt = a.w ? a.w[i] : 1.0;
        const double xx = a.xx[i];
        const int m = (int)a.mc[i];
        const double lk = sample_llk(xx, quad, Posterior<K>::logdet(pm, pe), s2, lnsig, m, K);
        double tr = 0.0;
        double *sc = a.sc + i * 4;
        if constexpr (EM) {
#pragma unroll
            for (int c = 0; c < K; ++c)  // w P = w (z z^T + s2 M^-1), packed
                tr += post.minv_column(c, [&](int r, int cc, double v) { g[tri(r, cc)] = wgt * (z[r] * z[cc] + s2 * v); });
#pragma unroll
            for (int c = 0; c < K; ++c) bz[c] = wgt * z[c];
            bz[K] = wgt;
            sc[0] = m > 0 ? wgt * s2 * ((double)K - s2 * tr) : 0.0;
            sc[1] = m > 0 ? wgt * (xx - quad - s2 * zz) : 0.0;
            sc[2] = wgt * lk;
            sc[3] = m > 0 ? 1.0 : 0.0;
        } else {
#pragma unroll
            for (int c = 0; c < K; ++c) {
                bz[c] = z[c];  // unweighted state for the reconstruction pass
                if (a.states) a.states[i * K + c] = z[c];
            }
            if (a.need_sigma) {
#pragma unroll
            for (int c = 0; c < K; ++c)
                (void)post.minv_column(c, [&](int r, int cc, double v) {
                    const double sv = s2 * v;
                    g[tri(r, cc)] = sv;  // Sigma packed, for covariance diagonals
                    if (a.covs) {
                        a.covs[(i * K + r) * K + cc] = sv;
                        a.covs[(i * K + cc) * K + r] = sv;
                    }
                });
            }
            sc[0] = 0.0;
            sc[1] = 0.0;
            sc[2] = wgt * lk;
            sc[3] = 0.0;
            if (a.llks) a.llks[i] = lk;
        }
    }
}
template <int K, bool EM>
__global__ __launch_bounds__(256) void solve_lane_kernel(SolveArgs a) {
    solve_lane_body<K, EM>(a);
}
// k >= 13: the packed factor alone is 182..272 registers -- one wave per SIMD with the whole 512-entry file (the
// accumulation half takes what the 256 directly addressable registers cannot hold; without the attribute: 148 dwords of
// scratch per lane at k = 16)
template <int K, bool EM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void solve_lane_wide_kernel(SolveArgs a) {
    solve_lane_body<K, EM>(a);
}
template <int K>
static hipError_t launch_solve_lane(const SolveArgs &a, int n_cu, hipStream_t s) {
    int64_t blocks = (a.n + 255) / 256;
    if (blocks > 8 * (int64_t)n_cu) blocks = 8 * (int64_t)n_cu;
    if (blocks < 1) blocks = 1;
    if constexpr (K >= 13) {
        if (a.em) hipLaunchKernelGGL((solve_lane_wide_kernel<K, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((solve_lane_wide_kernel<K, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else {
        if (a.em) hipLaunchKernelGGL((solve_lane_kernel<K, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((solve_lane_kernel<K, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ register-resident per-sample solve
// One wave per sample, no LDS: lane r keeps row r of M = G + s2 I (padded to KPAD with an identity block)
// in registers, lane c keeps column c of M^-1.  Every quantity the lanes share -- pivots, the L_cp of the
// trailing update, the L_at of the triangular solves, b_a, z_a -- is wave-uniform and is broadcast with
// v_readlane into a scalar operand of the FMA, so the O(k^3) loops are straight-line register code.
template <int KPAD>
__global__ __launch_bounds__(256) void solve_reg_kernel(SolveArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = a.k, kp = k * (k + 1) / 2;
    const double s2 = a.model[1], lnsig = a.model[2];
    auto bcast = [&](double v, int src) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < a.n; i += stride) {
        double *g = a.G + i * kp;
        double *bz = a.Bz + i * (k + 1);
        const bool live = lane < k;
        double row[KPAD];
        {
            const double *grow = g + (live ? lane * (lane + 1) / 2 : 0);
#pragma unroll
            for (int t = 0; t < KPAD; ++t) {
                double v = (t == lane) ? 1.0 : 0.0;            // identity padding
                if (live && t <= lane) v = grow[t] + (t == lane ? s2 : 0.0);
                row[t] = v;                                      // entries above the diagonal are never read
            }
        }
        const double bv = live ? bz[lane] : 0.0;
        // Cholesky, right-looking; diagonal slots keep 1 / L_pp
        double mant = 1.0;
        int ex = 0;
#pragma unroll
        for (int p = 0; p < KPAD; ++p) {
            const double piv = bcast(row[p], p);
            const double rinv = 1.0 / sqrt(piv);
            int e;
            mant *= frexp(piv, &e);
            ex += e;
            row[p] = (lane == p) ? rinv : row[p] * rinv;
#pragma unroll
            for (int c = p + 1; c < KPAD; ++c) row[c] -= row[p] * bcast(row[p], c);  // lanes < c touch unused slots
        }
        const double logdet = log(mant) + (double)ex * LN_2;
        // column `lane` of M^-1: L u = e_lane, L^T x = u; L_at is uniform -> scalar operand
        double u[KPAD];
#pragma unroll
        for (int r = 0; r < KPAD; ++r) {
            double sacc = (r == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int t = 0; t < r; ++t) sacc -= bcast(row[t], r) * u[t];
            u[r] = sacc * bcast(row[r], r);
        }
#pragma unroll
        for (int r = KPAD - 1; r >= 0; --r) {
            double sacc = u[r];
#pragma unroll
            for (int t = r + 1; t < KPAD; ++t) sacc -= bcast(row[r], t) * u[t];
            u[r] = sacc * bcast(row[r], r);
        }
        // z_c = sum_a (M^-1)_{ac} b_a (symmetry: lane c holds column c), quad = b^T z
        double z = 0.0, diag = 0.0;
#pragma unroll
        for (int r = 0; r < KPAD; ++r) {
            z += u[r] * bcast(bv, r);
            diag = (r == lane) ? u[r] : diag;
        }
        const double quad = gwave_sum(live ? bv * z : 0.0);
        const double zz = gwave_sum(live ? z * z : 0.0);
        const double tr = gwave_sum(live ? diag : 0.0);
        const double wgt = a.w ? a.w[i] : 1.0;
        const double xx = a.xx[i];
        const int m = (int)a.mc[i];
        const double lk = sample_llk(xx, quad, logdet, s2, lnsig, m, k);
        if (a.em) {
#pragma unroll
            for (int r = 0; r < KPAD; ++r) {
                const double zr = bcast(z, r);
                if (live && r >= lane && r < k) g[r * (r + 1) / 2 + lane] = wgt * (zr * z + s2 * u[r]);
            }
            if (live) bz[lane] = wgt * z;
            if (lane == 0) {
                bz[k] = wgt;
                double *sc = a.sc + i * 4;
                sc[0] = m > 0 ? wgt * s2 * ((double)k - s2 * tr) : 0.0;
                sc[1] = m > 0 ? wgt * (xx - quad - s2 * zz) : 0.0;
                sc[2] = wgt * lk;
                sc[3] = m > 0 ? 1.0 : 0.0;
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPAD; ++r) {
                const double sv = s2 * u[r];
                if (live && r < k) {
                    if (a.covs) a.covs[(i * k + r) * k + lane] = sv;
                    if (r >= lane) g[r * (r + 1) / 2 + lane] = sv;  // Sigma packed, for covariance diagonals
                }
            }
            if (live) {
                bz[lane] = z;  // unweighted state for the reconstruction pass
                if (a.states) a.states[i * k + lane] = z;
            }
            if (lane == 0) {
                double *sc = a.sc + i * 4;
                sc[0] = 0.0;
                sc[1] = 0.0;
                sc[2] = wgt * lk;
                sc[3] = 0.0;
                if (a.llks) a.llks[i] = lk;
            }
        }
    }
}


// ------------------------------------------------------------------ per-sample solve, LDS-broadcast form
// As solve_reg_kernel (lane r keeps row r of M in registers, lane c column c of M^-1), but the wave-uniform operand
// of every multiply-add -- L_cp in the trailing update, L_rt in the triangular solves -- comes from ONE uniform LDS
// read (all lanes, same address: a broadcast) of the factor the lanes publish column by column, instead of two
// v_readlane + the SGPR hazard wait per operand: the O(k^3) loops become "ds_read + v_fma" pairs that the LDS pipe
// and the fp64 pipe run side by side.  A wave's LDS operations execute in order, so a column is readable right after
// it is stored; no barrier.
// op i of a flattened column-oriented substitution on a KxK factor: unknown t scaled (r == t) or unknown r updated
// with unknown t; (lt, lr) = where its factor entry lives in Lm
struct SubOp {
    int t, r, lt, lr;
};
__host__ __device__ constexpr SubOp sub_op(int K, int i, bool fwd) {
    if (fwd) {
        int t = 0;
        while (i >= K - t) {  // column t has 1 + (K - 1 - t) ops
            i -= K - t;
            ++t;
        }
        return SubOp{t, t + i, t, t + i};
    }
    int t = K - 1;
    while (i >= 1 + t) {  // column t has 1 + t ops
        i -= 1 + t;
        --t;
    }
    if (i == 0) return SubOp{t, t, t, t};
    return SubOp{t, i - 1, i - 1, t};
}

template <int KPAD>
__global__ __launch_bounds__(256) void solve_bc_kernel(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *Lm = gsm + (size_t)wave * KPAD * KPAD;  // Lm[p][r] = L_rp (p <= r); Lm[p][p] = 1 / L_pp
    const int k = a.k, kp = k * (k + 1) / 2;
    const double s2 = a.model[1], lnsig = a.model[2];
    auto bcast = [&](double v, int src) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < a.n; i += stride) {
        double *g = a.G + i * kp;
        double *bz = a.Bz + i * (k + 1);
        const bool live = lane < k;
        double row[KPAD];
        {
            const double *grow = g + (live ? lane * (lane + 1) / 2 : 0);
#pragma unroll
            for (int t = 0; t < KPAD; ++t) {
                double v = (t == lane) ? 1.0 : 0.0;            // identity padding
                if (live && t <= lane) v = grow[t] + (t == lane ? s2 : 0.0);
                row[t] = v;
            }
        }
        const double bv = live ? bz[lane] : 0.0;
        double mant = 1.0;
        int ex = 0;
        constexpr int CH = 8;
        static_for_g<KPAD>([&](auto p_tag) {
            constexpr int p = decltype(p_tag)::value;
            const double piv = bcast(row[p], p);
            const double rinv = 1.0 / sqrt(piv);
            int e;
            mant *= frexp(piv, &e);
            ex += e;
            row[p] = (lane == p) ? rinv : row[p] * rinv;
            if (lane < KPAD) Lm[p * KPAD + lane] = row[p];
            // row[c] -= L_lane,p * L_cp for c > p (lanes < c touch unused slots); the uniform operands arrive in groups
            // of CH, the next group requested before the current one is consumed (fenced: hipcc otherwise hoists
            // every read of the kernel to the top and spills them)
            constexpr int NC = KPAD - 1 - p;
            if constexpr (NC > 0) {
                constexpr int NGC = (NC + CH - 1) / CH;
                double bc[3][CH];
                auto request = [&](auto g_tag) {
                    constexpr int g = decltype(g_tag)::value;
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const int o = g * CH + q;
                        bc[g % 3][q] = Lm[p * KPAD + p + 1 + (o < NC ? o : NC - 1)];
                    }
                };
                request(std::integral_constant<int, 0>{});
                if constexpr (NGC > 1) request(std::integral_constant<int, 1>{});
                static_for_g<NGC>([&](auto g_tag) {
                    constexpr int g = decltype(g_tag)::value;
                    if constexpr (g + 2 < NGC) request(std::integral_constant<int, g + 2>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const int o = g * CH + q;
                        if (o < NC) row[p + 1 + o] -= row[p] * bc[g % 3][q];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
        const double logdet = log(mant) + (double)ex * LN_2;
        // column `lane` of M^-1: L u = e_lane, then L^T x = u
        // column-oriented substitutions (once an unknown is final, the updates of all the others are independent),
        // each flattened into ONE stream of (uniform LDS operand, multiply-add) pairs -- the factor is static by now, so
        // the operand requests run a fixed distance (two groups of GP) ahead of their use across column boundaries:
        //   forward:  for t = 0 ..:      u_t *= 1/L_tt;  u_r -= L_rt u_t  (r > t)     operands Lm[t][t], Lm[t][r]
        //   backward: for t = K-1 .. 0:  u_t *= 1/L_tt;  u_r -= L_tr u_t  (r < t)     operands Lm[t][t], Lm[r][t]
        double u[KPAD];
#pragma unroll
        for (int r = 0; r < KPAD; ++r) u[r] = (r == lane) ? 1.0 : 0.0;
        constexpr int NOPS = KPAD + KPAD * (KPAD - 1) / 2, GP = 8, NG = (NOPS + GP - 1) / GP;
        auto stream = [&](auto fwd_tag) {
            constexpr bool FWD = decltype(fwd_tag)::value;
            double ring[3][GP];
            auto request = [&](auto g_tag) {
                constexpr int g = decltype(g_tag)::value;
                static_for_g<GP>([&](auto q_tag) {
                    constexpr int i = g * GP + decltype(q_tag)::value;
                    constexpr SubOp op = sub_op(KPAD, i < NOPS ? i : NOPS - 1, FWD);
                    ring[g % 3][decltype(q_tag)::value] = Lm[op.lt * KPAD + op.lr];
                });
            };
            request(std::integral_constant<int, 0>{});
            if constexpr (NG > 1) request(std::integral_constant<int, 1>{});
            static_for_g<NG>([&](auto g_tag) {
                constexpr int g = decltype(g_tag)::value;
                if constexpr (g + 2 < NG) request(std::integral_constant<int, g + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for_g<GP>([&](auto q_tag) {
                    constexpr int i = g * GP + decltype(q_tag)::value;
                    if constexpr (i < NOPS) {
                        constexpr SubOp op = sub_op(KPAD, i, FWD);
                        const double v = ring[g % 3][decltype(q_tag)::value];
                        if constexpr (op.r == op.t) u[op.t] *= v;
                        else u[op.r] -= v * u[op.t];
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        stream(std::true_type{});
        stream(std::false_type{});
        double z = 0.0, diag = 0.0;
#pragma unroll
        for (int r = 0; r < KPAD; ++r) {
            z += u[r] * bcast(bv, r);
            diag = (r == lane) ? u[r] : diag;
        }
        const double quad = gwave_sum(live ? bv * z : 0.0);
        const double zz = gwave_sum(live ? z * z : 0.0);
        const double tr = gwave_sum(live ? diag : 0.0);
        const double wgt = a.w ? a.w[i] : 1.0;
        const double xx = a.xx[i];
        const int m = (int)a.mc[i];
        const double lk = sample_llk(xx, quad, logdet, s2, lnsig, m, k);
        if (a.em) {
#pragma unroll
            for (int r = 0; r < KPAD; ++r) {
                const double zr = bcast(z, r);
                if (live && r >= lane && r < k) g[r * (r + 1) / 2 + lane] = wgt * (zr * z + s2 * u[r]);
            }
            if (live) bz[lane] = wgt * z;
            if (lane == 0) {
                bz[k] = wgt;
                double *sc = a.sc + i * 4;
                sc[0] = m > 0 ? wgt * s2 * ((double)k - s2 * tr) : 0.0;
                sc[1] = m > 0 ? wgt * (xx - quad - s2 * zz) : 0.0;
                sc[2] = wgt * lk;
                sc[3] = m > 0 ? 1.0 : 0.0;
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPAD; ++r) {
                const double sv = s2 * u[r];
                if (live && r < k) {
                    if (a.covs) a.covs[(i * k + r) * k + lane] = sv;
                    if (r >= lane) g[r * (r + 1) / 2 + lane] = sv;
                }
            }
            if (live) {
                bz[lane] = z;
                if (a.states) a.states[i * k + lane] = z;
            }
            if (lane == 0) {
                double *sc = a.sc + i * 4;
                sc[0] = 0.0;
                sc[1] = 0.0;
                sc[2] = wgt * lk;
                sc[3] = 0.0;
                if (a.llks) a.llks[i] = lk;
            }
        }
    }
}

template <int KPAD>
static hipError_t launch_solve_bc(const SolveArgs &a, int grid, hipStream_t s) {
    const size_t lds = sizeof(double) * 4 * KPAD * KPAD;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (lds > 65536 && !(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_bc_kernel<KPAD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((solve_bc_kernel<KPAD>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}


// ------------------------------------------------------------------ per-sample solve on the fp64 MFMA (blocked)
// One wave per sample, M = G + s2 I (padded with an identity block to n = 16 NB) in the wave's LDS area, inverted IN
// PLACE by the three classical blocked sweeps over 16 x 16 blocks -- every block product is four v_mfma_f64_16x16x4:
//   potrf   for j: L_jj = chol(A_jj), T_jj = L_jj^-1 (kept in the diagonal block); L_ij = A_ij T_jj^T (i > j);
//           A_il -= L_ij L_lj^T (j < l <= i)
//   trtri   W = L^-1: for j descending, i descending: W_ij = -(sum_{t=j+1..i} W_it L_tj) T_jj      (W_ii = T_ii)
//   lauum   M^-1 = W^T W: for i, for l <= i (l = i last): R_il = sum_{t >= i} W_ti^T W_tl
// 52 block products at NB = 4 (13 k MFMA cycles) against ~6 k dependent v_readlane / LDS-broadcast multiply-adds per
// sample of the lane-per-row forms above.  The 16 x 16 diagonal blocks are factored and inverted by the lanes
// themselves (lane = row, four redundant copies, uniform LDS reads for the shared operands).  Outputs leave through
// the packed index (coalesced), z = M^-1 b by symmetric row reads.
// (round 6) NB = 5 .. 8: state sizes 65 .. 128 (the reference bounds k nowhere, ppca_model.rs:51-70).  Round 5 gave them one
// WORKGROUP per matrix with a __syncthreads() per column (solve_big_kernel: 1.2 s per million systems at k = 65); the blocked form
// needs nothing but LDS -- 36 KB (NB = 5) to 76 KB (NB = 8: blocks unpadded) per sample, so 4 / 3 / 2 / 2 waves per CU.
constexpr int solve_ld(int nb) { return nb == 8 ? 16 : 18; }
constexpr int solve_mfma_waves(int nb) { return nb <= 3 ? 4 : nb == 4 ? 3 : nb == 5 ? 4 : nb == 6 ? 3 : 2; }
template <int NB>
__device__ __forceinline__ void solve_mfma_body(const SolveArgs &a) {
    // the lower block triangle only, block after block (row stride LD inside a 16 x 16 block): 24 KB per sample at NB = 4
    // instead of 34, so that two workgroups (of three waves there) share a CU -- two waves per SIMD
    constexpr int N = 16 * NB, LD = solve_ld(NB), BSZ = 16 * LD, NBK = NB * (NB + 1) / 2, W = solve_mfma_waves(NB);
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int l15 = lane & 15, l4 = lane >> 4;
    double *Am = gsm + (size_t)wave * (NBK * BSZ + 2 * N);
    double *zs = Am + NBK * BSZ;   // z (N)
    double *bs = zs + N;        // b (N)
    const int k = a.k, kp = k * (k + 1) / 2;
    const double s2 = a.model[1], lnsig = a.model[2];
    auto bcast = [&](double v, int src) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    auto blk = [&](int i, int j) { return Am + (i * (i + 1) / 2 + j) * BSZ; };  // i >= j
    auto at = [&](int r, int c) { return blk(r >> 4, c >> 4) + (r & 15) * LD + (c & 15); };  // block row >= block column
    // D = sign * X' Y' + C with X' = X or X^T, Y' = Y or Y^T (16 x 16 blocks in LDS, leading dimension LD)
    auto mma = [&](d4g_t acc, const double *X, bool xT, const double *Y, bool yT, double sign) {
        __builtin_amdgcn_sched_barrier(0);  // (one block product's operands in flight at a time: the unrolled sweeps hoisted dozens)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double av = xT ? X[(4 * s + l4) * LD + l15] : X[l15 * LD + 4 * s + l4];
            const double bv = yT ? Y[l15 * LD + 4 * s + l4] : Y[(4 * s + l4) * LD + l15];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * av, bv, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        return acc;
    };
    auto ldC = [&](const double *Z) {
        d4g_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = Z[(l4 + 4 * r) * LD + l15];
        return v;
    };
    auto stC = [&](double *Z, d4g_t v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[(l4 + 4 * r) * LD + l15] = v[r];
    };
    if (wave >= W) return;
    const int64_t stride = (int64_t)gridDim.x * W;
    for (int64_t i = (int64_t)blockIdx.x * W + wave; i < a.n; i += stride) {
        // (the lane index goes through an opaque statement per sample: the per-lane LDS addresses of the phases below are then
        //  recomputed where they are used instead of being hoisted out of the loop and parked -- in scratch at 256 registers)
        asm volatile("" : "+v"(lane));
        l15 = lane & 15;
        l4 = lane >> 4;
        double *g = a.G + i * kp;
        double *bz = a.Bz + i * (k + 1);
        // ---- M into LDS, full symmetric, identity padding; b.  The packed Gram is requested in one burst (NPK loads
        // in flight per lane) before the first LDS store; each lane walks the packed index by 64 (no square roots).
        constexpr int NPK = (N * (N + 1) / 2 + 63) / 64;
        constexpr int NPC = NPK <= 33 ? NPK : 32;  // loads in flight per lane (NB > 4: bursts of 32 -- 129 at once would be 258 registers)
        if (k < N) {
            for (int e = lane; e < NBK * BSZ; e += 64) Am[e] = 0.0;
            for (int e = lane; e < N; e += 64)
                if (e >= k) *at(e, e) = 1.0;
        }
        for (int e = lane; e < N; e += 64) bs[e] = (e < k) ? bz[e] : 0.0;
        {
            int r = (int)((sqrt(8.0 * (double)lane + 1.0) - 1.0) * 0.5);
            while ((r + 1) * (r + 2) / 2 <= lane) ++r;
            while (r * (r + 1) / 2 > lane) --r;
            int c = lane - r * (r + 1) / 2;
            for (int q0 = 0; q0 < NPK; q0 += NPC) {
                double gv[NPC];
#pragma unroll
                for (int q = 0; q < NPC; ++q) {
                    const int e = lane + 64 * (q0 + q);
                    gv[q] = g[e < kp ? e : kp - 1];
                }
#pragma unroll
                for (int q = 0; q < NPC; ++q) {
                    if (lane + 64 * (q0 + q) < kp) {
                        const double v = gv[q] + (r == c ? s2 : 0.0);
                        *at(r, c) = v;
                        if ((r >> 4) == (c >> 4)) *at(c, r) = v;  // (diagonal blocks are kept whole: their factorisation reads rows)
                    }
                    c += 64;
                    while (c > r) {
                        c -= r + 1;
                        ++r;
                    }
                }
            }
        }
        double mant = 1.0;
        int ex = 0;
        // ---- potrf
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double *D = blk(j, j);
            {   // diagonal block: L (lane = row l15), then T = L^-1 (lane = column l15), written back with a zero upper part
                // (tried: the entries of L by DPP row_newbcast from the owning lane's register instead of uniform LDS reads --
                //  376 reads per block fewer, but the unrolled broadcasts went to scratch at 256 registers: 306 against 216 ms)
                double row[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) row[c] = D[l15 * LD + c];
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const double piv = bcast(row[p], p);
                    const double rinv = fast_rsqrt(piv);
                    int e;
                    mant *= frexp(piv, &e);
                    ex += e;
                    row[p] = (l15 == p) ? rinv : row[p] * rinv;
                    D[l15 * LD + p] = row[p];  // column p of L (diagonal slot: 1 / L_pp); the four copies agree
#pragma unroll
                    for (int c = p + 1; c < 16; ++c) row[c] -= row[p] * D[c * LD + p];
                }
                double u[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) u[r] = (r == l15) ? 1.0 : 0.0;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    u[t] *= D[t * LD + t];
#pragma unroll
                    for (int r = t + 1; r < 16; ++r) u[r] -= D[r * LD + t] * u[t];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) D[r * LD + l15] = (r >= l15) ? u[r] : 0.0;
            }
#pragma unroll
            for (int ii = j + 1; ii < NB; ++ii) {  // L_ij = A_ij T_jj^T
                d4g_t acc = {0, 0, 0, 0};
                acc = mma(acc, blk(ii, j), false, D, true, 1.0);
                stC(blk(ii, j), acc);
            }
#pragma unroll
            for (int ii = j + 1; ii < NB; ++ii)
#pragma unroll
                for (int l = j + 1; l <= ii; ++l) {  // A_il -= L_ij L_lj^T
                    d4g_t acc = ldC(blk(ii, l));
                    acc = mma(acc, blk(ii, j), false, blk(l, j), true, -1.0);
                    stC(blk(ii, l), acc);
                }
        }
        const double logdet = log(mant) + (double)ex * LN_2;
        // ---- trtri: W = L^-1 in place (diagonal blocks already hold T)
#pragma unroll
        for (int j = NB - 2; j >= 0; --j)
#pragma unroll
            for (int ii = NB - 1; ii > j; --ii) {
                d4g_t acc = {0, 0, 0, 0};
#pragma unroll
                for (int t = j + 1; t <= ii; ++t) acc = mma(acc, blk(ii, t), false, blk(t, j), false, 1.0);
                stC(blk(ii, j), acc);  // Y_i (L_ij is no longer needed: rows above use L_tj with t < i only)
                d4g_t w = {0, 0, 0, 0};
                w = mma(w, blk(ii, j), false, blk(j, j), false, -1.0);
                stC(blk(ii, j), w);
            }
        // ---- lauum: M^-1 = W^T W, lower blocks, in place
#pragma unroll
        for (int ii = 0; ii < NB; ++ii)
#pragma unroll
            for (int l = 0; l <= ii; ++l) {
                d4g_t acc = {0, 0, 0, 0};
#pragma unroll
                for (int t = ii; t < NB; ++t) acc = mma(acc, blk(t, ii), true, blk(t, l), false, 1.0);
                stC(blk(ii, l), acc);
            }
        // ---- z = M^-1 b (lane = row; the strict upper part is read through the symmetric entry), traces
        double zv[(N + 63) / 64];
        double quad = 0.0, zz = 0.0, tr = 0.0;
#pragma unroll
        for (int q = 0; q < (N + 63) / 64; ++q) {
            const int r = lane + 64 * q;
            double zacc = 0.0;
            if (r < N) {
                for (int c = 0; c < N; ++c) {
                    const double mv = (c <= r) ? *at(r, c) : *at(c, r);
                    zacc += mv * bs[c];
                }
                zs[r] = zacc;
                if (r < k) {
                    quad += bs[r] * zacc;
                    zz += zacc * zacc;
                    tr += *at(r, r);
                }
            }
            zv[q] = zacc;
        }
        quad = gwave_sum(quad);
        zz = gwave_sum(zz);
        tr = gwave_sum(tr);
        const double wgt = a.w ? a.w[i] : 1.0;
        const double xx = a.xx[i];
        const int m = (int)a.mc[i];
        const double lk = sample_llk(xx, quad, logdet, s2, lnsig, m, k);
        if (a.em) {
            // w P = w (z z^T + s2 M^-1), packed, by packed index (coalesced)
            {
                int r = (int)((sqrt(8.0 * (double)lane + 1.0) - 1.0) * 0.5);
                while ((r + 1) * (r + 2) / 2 <= lane) ++r;
                while (r * (r + 1) / 2 > lane) --r;
                int c = lane - r * (r + 1) / 2;
#pragma unroll
                for (int q = 0; q < NPK; ++q) {
                    const int e = lane + 64 * q;
                    if (e < kp) g[e] = wgt * (zs[r] * zs[c] + s2 * *at(r, c));
                    c += 64;
                    while (c > r) {
                        c -= r + 1;
                        ++r;
                    }
                }
            }
            if (lane < k) bz[lane] = wgt * zv[0];
            if constexpr (N > 64) {
                if (lane + 64 < k) bz[lane + 64] = wgt * zv[1];
            }
            if (lane == 0) {
                bz[k] = wgt;
                double *sc = a.sc + i * 4;
                sc[0] = m > 0 ? wgt * s2 * ((double)k - s2 * tr) : 0.0;
                sc[1] = m > 0 ? wgt * (xx - quad - s2 * zz) : 0.0;
                sc[2] = wgt * lk;
                sc[3] = m > 0 ? 1.0 : 0.0;
            }
        } else {
            for (int e = lane; e < kp; e += 64) {
                int r = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while ((r + 1) * (r + 2) / 2 <= e) ++r;
                while (r * (r + 1) / 2 > e) --r;
                const int c = e - r * (r + 1) / 2;
                g[e] = s2 * *at(r, c);  // Sigma packed, for the covariance diagonals
            }
            if (a.covs) {
                for (int e = lane; e < k * k; e += 64) {
                    const int r = e / k, c = e - r * k;
                    a.covs[i * (int64_t)k * k + e] = s2 * (c <= r ? *at(r, c) : *at(c, r));
                }
            }
            if (lane < k) {
                bz[lane] = zv[0];
                if (a.states) a.states[i * k + lane] = zv[0];
            }
            if constexpr (N > 64) {
                if (lane + 64 < k) {
                    bz[lane + 64] = zv[1];
                    if (a.states) a.states[i * k + lane + 64] = zv[1];
                }
            }
            if (lane == 0) {
                double *sc = a.sc + i * 4;
                sc[0] = 0.0;
                sc[1] = 0.0;
                sc[2] = wgt * lk;
                sc[3] = 0.0;
                if (a.llks) a.llks[i] = lk;
            }
        }
        if constexpr (N > 64) {  // (bz of the sample is read in the image phase and written here through lanes of ONE wave: ordered)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void solve_mfma_kernel(SolveArgs a) {
    solve_mfma_body<NB>(a);
}
// the same capped at 256 registers: two workgroups per CU, two waves per SIMD (the default)
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void solve_mfma_occ2_kernel(SolveArgs a) {
    solve_mfma_body<NB>(a);
}

template <int NB>
static hipError_t launch_solve_mfma(const SolveArgs &a, int n_cu, hipStream_t s) {
    constexpr int N = 16 * NB, W = solve_mfma_waves(NB);
    const size_t lds = sizeof(double) * W * (NB * (NB + 1) / 2 * 16 * solve_ld(NB) + 2 * N);
    // PPCA_SOLVE_OCC2=0: one workgroup per CU with the whole register file (A/B runs; the form of rounds 2-3)
    static const bool occ2 = NB <= 4 && [] {
        const char *e = getenv("PPCA_SOLVE_OCC2");
        return !(e && atoi(e) == 0);
    }();
    int grid = (int)std::min<int64_t>((a.n + W - 1) / W, (int64_t)n_cu * (occ2 ? 2 : 1));
    if (grid < 1) grid = 1;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (lds > 65536 && !(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_mfma_kernel<NB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if constexpr (NB <= 4) {
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_mfma_occ2_kernel<NB>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    if constexpr (NB <= 4) {
        if (occ2) {
            hipLaunchKernelGGL((solve_mfma_occ2_kernel<NB>), dim3(grid), dim3(64 * W), lds, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((solve_mfma_kernel<NB>), dim3(grid), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}
static hipError_t launch_solve(const SolveArgs &a, int n_cu, hipStream_t s) {
    if (a.k > 64) {
        static const bool big = [] {  // PPCA_SOLVE_BIG=1: the workgroup-per-matrix solver of round 5 (A/B runs)
            const char *e = getenv("PPCA_SOLVE_BIG");
            return e && atoi(e) == 1;
        }();
        if (big) return launch_solve_big(a, n_cu, s);
        if (a.k <= 80) return launch_solve_mfma<5>(a, n_cu, s);
        if (a.k <= 96) return launch_solve_mfma<6>(a, n_cu, s);
        if (a.k <= 112) return launch_solve_mfma<7>(a, n_cu, s);
        return launch_solve_mfma<8>(a, n_cu, s);
    }
    int grid = (int)std::min<int64_t>((a.n + 3) / 4, (int64_t)n_cu);
    if (grid < 1) grid = 1;
    static const bool reg = [] {  // PPCA_GENERIC_REG_SOLVE=1: the v_readlane-broadcast form (A/B runs)
        const char *e = getenv("PPCA_GENERIC_REG_SOLVE");
        return e && atoi(e) == 1;
    }();
    static const int form = [] {  // PPCA_GENERIC_SOLVE = mfma (default) | bc (LDS-broadcast, lane per row)
        const char *e = getenv("PPCA_GENERIC_SOLVE");
        return (e && e[0] == 'b') ? 1 : 0;
    }();
    static const bool lane_ok = [] {  // PPCA_GENERIC_LANE_SOLVE=0: the wave-per-sample forms at every k (A/B runs)
        const char *e = getenv("PPCA_GENERIC_LANE_SOLVE");
        return !(e && atoi(e) == 0);
    }();
    if (lane_ok && a.k <= 16) {
        switch (a.k) {
#define PPCA_LANE_CASE(KK) \
    case KK:               \
        return launch_solve_lane<KK>(a, n_cu, s);
            PPCA_LANE_CASE(1) PPCA_LANE_CASE(2) PPCA_LANE_CASE(3) PPCA_LANE_CASE(4) PPCA_LANE_CASE(5) PPCA_LANE_CASE(6)
            PPCA_LANE_CASE(7) PPCA_LANE_CASE(8) PPCA_LANE_CASE(9) PPCA_LANE_CASE(10) PPCA_LANE_CASE(11) PPCA_LANE_CASE(12)
            PPCA_LANE_CASE(13) PPCA_LANE_CASE(14) PPCA_LANE_CASE(15) PPCA_LANE_CASE(16)
#undef PPCA_LANE_CASE
        }
    }
    static const bool batched = [] {  // PPCA_SOLVE4=0: one sample per wave (solve_mfma_body; the form of rounds 2-4, A/B runs)
        const char *e = getenv("PPCA_SOLVE4");
        return !(e && atoi(e) == 0);
    }();
    if (!reg && form == 0 && batched && solve4_covers(a.k)) return launch_solve4(a, n_cu, s);
    if (!reg && form == 0) {
        if (a.k <= 16) return launch_solve_mfma<1>(a, n_cu, s);
        if (a.k <= 32) return launch_solve_mfma<2>(a, n_cu, s);
        if (a.k <= 48) return launch_solve_mfma<3>(a, n_cu, s);
        return launch_solve_mfma<4>(a, n_cu, s);
    }
    if (!reg) {
        if (a.k <= 16) return launch_solve_bc<16>(a, grid, s);
        if (a.k <= 32) return launch_solve_bc<32>(a, grid, s);
        return launch_solve_bc<64>(a, grid, s);
    }
    if (a.k <= 16) hipLaunchKernelGGL((solve_reg_kernel<16>), dim3(grid), dim3(256), 0, s, a);
    else if (a.k <= 32) hipLaunchKernelGGL((solve_reg_kernel<32>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((solve_reg_kernel<64>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// strided column sum of sc[n][4] (+ sum of weights) into scal[8]; one block, deterministic
// Two stages, both in a fixed order (deterministic): blocks of SCAL_ROWS samples -> part[block][5], then one workgroup
// over the partials.  (One workgroup over a whole chunk -- a million rows at small k -- took milliseconds per chunk.)
constexpr int SCAL_ROWS = 4096;
__global__ __launch_bounds__(256) void scal_reduce_kernel(const double *sc, const double *w, int64_t n, double *part) {
    __shared__ double red[256][5];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * SCAL_ROWS, i1 = i0 + SCAL_ROWS < n ? i0 + SCAL_ROWS : n;
    double v[5] = {0, 0, 0, 0, 0};
    for (int64_t i = i0 + tid; i < i1; i += 256) {
        v[0] += sc[i * 4 + 0];
        v[1] += sc[i * 4 + 1];
        v[2] += sc[i * 4 + 2];
        v[3] += sc[i * 4 + 3];
        v[4] += w ? w[i] : 1.0;
    }
    for (int c = 0; c < 5; ++c) red[tid][c] = v[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
            for (int c = 0; c < 5; ++c) red[tid][c] += red[tid + o][c];
        __syncthreads();
    }
    if (tid < 5) part[(int64_t)blockIdx.x * 5 + tid] = red[0][tid];
}
__global__ __launch_bounds__(256) void scal_final_kernel(const double *part, int nblocks, double *scal, int accumulate) {
    __shared__ double red[256][5];
    const int tid = threadIdx.x;
    double v[5] = {0, 0, 0, 0, 0};
    for (int b = tid; b < nblocks; b += 256)
        for (int c = 0; c < 5; ++c) v[c] += part[(int64_t)b * 5 + c];
    for (int c = 0; c < 5; ++c) red[tid][c] = v[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
            for (int c = 0; c < 5; ++c) red[tid][c] += red[tid + o][c];
        __syncthreads();
    }
    if (tid == 0) {
        const double r[8] = {red[0][0], red[0][1], red[0][2], red[0][4], red[0][3], 0.0, 0.0, 0.0};
        for (int c = 0; c < 8; ++c) scal[c] = accumulate ? scal[c] + r[c] : r[c];
    }
}

// reconstruction / covariance diagonal from states (Bz, ld k+1) and packed Sigma (G)
__global__ void recon_kernel(const double *X, int64_t ldx, int64_t n, int d, int k, const double *model,
                             const double *Bz, const double *G, int mode, double *out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const int64_t i = idx / d;
    const int j = (int)(idx - i * d);
    const double *c = model + MODEL_HDR + (int64_t)j * k;
    const double mean = model[MODEL_HDR + (int64_t)d * k + j];
    const double x = X[i * ldx + j];
    const bool obs = __builtin_isfinite(x);
    double o;
    if (mode <= 1) {
        double s = mean;
        const double *z = Bz + i * (k + 1);
        for (int a = 0; a < k; ++a) s += c[a] * z[a];
        o = (mode == 1 && obs) ? x : s;
    } else {
        const double *sg = G + i * (int64_t)(k * (k + 1) / 2);
        double v = 0.0;
        for (int a = 0; a < k; ++a) {
            double t = 0.0;
            for (int b = 0; b < k; ++b) t += sg[a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a] * c[b];
            v += c[a] * t;
        }
        o = v + model[1];
        if (mode == 3 && obs) o = 0.0;
    }
    out[idx] = o;
}

// The same outputs with the work laid out for the machine (round 3, late; recon_kernel above is kept for A/B runs,
// PPCA_GENERIC_RECON=naive): a thread owns one DIMENSION (its row of C in registers, zero-padded to KPAD columns) and
// walks the samples of its block; everything per sample -- the state z, the packed Sigma -- is wave-uniform and comes
// through scalar loads, so an element costs its multiply-adds and one coalesced store (plus the load of x where the
// mode looks at it: smooth does not).  The naive kernel re-read C, z and Sigma per ELEMENT and divided a 64-bit index:
// 14 ms for 2 M x 200 at k = 16 where the output is 1.6 ms of HBM; its covariance diagonal -- k^2 indexed loads of
// Sigma per element -- 227 ms.
template <int KPAD>
__global__ __launch_bounds__(256) void recon2_kernel(const double *__restrict__ X, int64_t ldx, int64_t n, int d, int k,
                                                     const double *__restrict__ model, const double *__restrict__ Bz,
                                                     const double *__restrict__ G, int mode, double *__restrict__ out,
                                                     int rows_per_block) {
    const int j = blockIdx.y * 256 + threadIdx.x;
    const bool jv = j < d;
    double c[KPAD];
#pragma unroll
    for (int a = 0; a < KPAD; ++a) c[a] = (jv && a < k) ? model[MODEL_HDR + (int64_t)j * k + a] : 0.0;
    const double mean = jv ? model[MODEL_HDR + (int64_t)d * k + j] : 0.0;
    const double s2 = model[1];
    const int kp = k * (k + 1) / 2;
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t i1 = i0 + rows_per_block < n ? i0 + rows_per_block : n;
    for (int64_t i = i0; i < i1; ++i) {
        double o;
        if (mode <= 1) {
            const double *z = Bz + i * (k + 1);
            double sacc = mean;
#pragma unroll
            for (int a = 0; a < KPAD; ++a) {
                const double za = a < k ? z[a] : 0.0;  // (uniform: a scalar load and select)
                sacc += c[a] * za;
            }
            o = sacc;
        } else {
            const double *sg = G + i * (int64_t)kp;
            double v = 0.0;
#pragma unroll
            for (int a = 0; a < KPAD; ++a) {
                if (a < k) {  // (uniform)
                    double t = 0.0;
#pragma unroll
                    for (int b = 0; b < a; ++b) t += sg[a * (a + 1) / 2 + b] * c[b];
                    v += c[a] * (sg[a * (a + 1) / 2 + a] * c[a] + 2.0 * t);
                }
            }
            o = v + s2;
        }
        if (mode == 1 || mode == 3) {
            const double x = jv ? X[i * ldx + j] : 0.0;
            const bool obs = __builtin_isfinite(x);
            if (obs) o = (mode == 1) ? x : 0.0;
        }
        if (jv) out[i * d + j] = o;
    }
}

// ------------------------------------------------------------------ finalisation (runtime k)
// One wave per dimension: S_j (+ tau I) c = cross_j by Cholesky in LDS; old row kept if not SPD.
__global__ __launch_bounds__(128) void gen_rowsolve_kernel(const double *stats, const double *min, double *mout, int d,
                                                           int k, double tau) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kp = k * (k + 1) / 2, LD = k | 1;
    double *Mw = gsm + (size_t)wave * (k * LD + 64);
    const StatsLayout L(d, k);
    for (int j = blockIdx.x * 2 + wave; j < d; j += gridDim.x * 2) {
        const double *S = stats + L.S + (int64_t)j * kp;
        wave_sync();
        for (int e = lane; e < kp; e += 64) {
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            int c = e - r * (r + 1) / 2;
            Mw[r * LD + c] = S[e] + (r == c ? tau : 0.0);
        }
        double logdet, quad;
        const bool ok = wave_cholesky(Mw, k, LD, lane, logdet);
        const double rhs = lane < k ? stats[L.cross + (int64_t)j * k + lane] : 0.0;
        const double sol = wave_chol_solve(Mw, k, LD, lane, rhs, quad);
        if (lane < k) {
            const double old = min[MODEL_HDR + (int64_t)j * k + lane];
            mout[MODEL_HDR + (int64_t)j * k + lane] = ok ? sol : old;  // :313-321 keep the old row
        }
    }
}

__global__ void gen_finalize_misc_kernel(const double *stats, const double *min, double *mout, int d, int k, int has_ig,
                                         double alpha, double beta) {
    __shared__ double red[256];
    const StatsLayout L(d, k);
    const int tid = threadIdx.x;
    double ts = 0.0;
    for (int j = tid; j < d; j += 256) ts += stats[L.totals + j];
    red[tid] = ts;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double totsum = red[0];
    // (clamp: see finalize_kernel, ppca_kernels.hip)
    const double sq = stats[L.scalars + SC_SQERR], dv = fmax(stats[L.scalars + SC_DEVSQ], -stats[L.scalars + SC_SQERR]);
    const double s2new = has_ig ? ((sq + dv) / 2.0 + beta) / (totsum / 2.0 + alpha + 1.0) : (sq + dv) / totsum;
    const double *Cold = min + MODEL_HDR;
    const double *Mold = Cold + (int64_t)d * k;
    double *Mnew = mout + MODEL_HDR + (int64_t)d * k;
    for (int j = tid; j < d; j += 256) {
        double cz = 0.0;
        for (int a = 0; a < k; ++a) cz += Cold[(int64_t)j * k + a] * stats[L.U + (int64_t)j * k + a];
        const double tot = stats[L.totals + j];
        Mnew[j] = (tot > 0.0 ? (stats[L.sumx + j] - cz) / tot : 0.0) + Mold[j];
    }
    if (tid == 0) {
        const double sig = sqrt(s2new);
        mout[0] = sig;
        mout[1] = sig * sig;
        mout[2] = log(sig);
        mout[3] = 0.0;
    }
}

// ------------------------------------------------------------------ host orchestration
static int64_t gen_chunk(int k) {
    const int64_t kp = (int64_t)k * (k + 1) / 2;
    int64_t c = (int64_t)(1.5e9 / (8.0 * (double)(kp + k + 8)));  // ~1.5 GB of per-sample workspace
    if (c > (1 << 20)) c = 1 << 20;
    if (c < 4096) c = 4096;
    if (const char *e = getenv("PPCA_GEN_CHUNK")) {  // tests: several chunks at small N (a multiple of 64)
        const int64_t v = atoll(e);
        if (v >= 64) c = v / 64 * 64;
    }
    return c;
}

// split-K scratch: up to 4 dense (d x k') partials (and >= 16 of the (d x (k+1)) ones)
static int64_t gen_part_doubles(int d, int k) {
    const int64_t kp = (int64_t)k * (k + 1) / 2;
    // (small shapes: room for up to 64 K-slices of the mask-side statistics contraction, capped at 64 MB)
    return std::max<int64_t>(std::max<int64_t>(4 * (int64_t)d * kp, 64 * (int64_t)d * (k + 1)),
                             std::min<int64_t>(64 * (int64_t)d * kp, (int64_t)8 << 20));
}

static int64_t pad64(int64_t v) { return (v + 63) / 64 * 64; }

struct GenWs {
    double *Q, *G, *Bz, *xx, *mc, *sc, *part;
    int64_t chunk, part_cap;
    // int8-sliced contractions
    unsigned char *A, *AT;
    signed char *BtQ, *BtW;
    double *scaleQ, *scaleW, *predW, *colpart, *rmin, *spart;
    int *flags, *redoW;
    int dpad;
    int64_t npad;
    // two-kernel EM pass for 11 <= k <= 16, d <= 256 (ppca_em16.hip): rows handed between the kernels, tile masks,
    // per-workgroup partial statistics, [160 scales | 16 guard flags + 1 | slice table]
    double *e16_W, *e16_part, *e16_q;
    unsigned *e16_Mb;
};
constexpr int E16_MAX_GRID = 512;
static size_t carve_impl(void *ws, int d, int k, int64_t n, GenWs *out) {
    const int64_t kp = (int64_t)k * (k + 1) / 2;
    GenWs w{};
    w.chunk = std::min<int64_t>(gen_chunk(k), std::max<int64_t>(n, 1));
    w.dpad = (int)pad64(d);
    w.npad = pad64(w.chunk);
    w.part_cap = gen_part_doubles(d, k);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = ws ? static_cast<char *>(ws) + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return p;
    };
    w.Q = static_cast<double *>(take(sizeof(double) * (size_t)d * kp));
    w.G = static_cast<double *>(take(sizeof(double) * (size_t)w.chunk * kp));
    w.Bz = static_cast<double *>(take(sizeof(double) * (size_t)w.chunk * (k + 1)));
    w.xx = static_cast<double *>(take(sizeof(double) * (size_t)w.chunk));
    w.mc = static_cast<double *>(take(sizeof(double) * (size_t)w.chunk));
    w.sc = static_cast<double *>(take(sizeof(double) * 4 * (size_t)w.chunk));
    w.part = static_cast<double *>(take(sizeof(double) * (size_t)w.part_cap));
    w.A = static_cast<unsigned char *>(take((size_t)w.npad * w.dpad));
    w.AT = static_cast<unsigned char *>(take((size_t)w.dpad * w.npad));
    w.BtQ = static_cast<signed char *>(take((size_t)GQS * kp * w.dpad));
    w.BtW = static_cast<signed char *>(take((size_t)GQS * kp * w.npad));
    w.scaleQ = static_cast<double *>(take(sizeof(double) * (size_t)kp));
    w.scaleW = static_cast<double *>(take(sizeof(double) * (size_t)kp));
    w.predW = static_cast<double *>(take(sizeof(double) * (size_t)kp));
    w.redoW = static_cast<int *>(take(sizeof(int) * (size_t)kp));
    w.colpart = static_cast<double *>(take(sizeof(double) * 2 * (size_t)kp * (size_t)((w.chunk + 255) / 256)));
    w.rmin = static_cast<double *>(take(256));
    w.spart = static_cast<double *>(take(sizeof(double) * 5 * (size_t)((w.chunk + SCAL_ROWS - 1) / SCAL_ROWS)));
    w.flags = static_cast<int *>(take(256));
    if (em16_covers(d, k)) {
        const StatsLayout L(d, k);
        w.e16_W = static_cast<double *>(take(sizeof(double) * (size_t)w.chunk * em16_ncol(k)));
        w.e16_Mb = static_cast<unsigned *>(take(sizeof(unsigned) * 256 * (size_t)((w.chunk + 31) / 32)));
        w.e16_part = static_cast<double *>(take(sizeof(double) * (size_t)E16_MAX_GRID * L.len));
        w.e16_q = static_cast<double *>(take(sizeof(double) * 192 + em16_qtab_bytes(k)));
    }
    if (out) *out = w;
    return off;
}
size_t generic_workspace_bytes(int d, int k, int64_t n) { return carve_impl(nullptr, d, k, n, nullptr); }
static GenWs carve(void *ws, int d, int k, int64_t n) {
    GenWs w;
    (void)carve_impl(ws, d, k, n, &w);
    return w;
}

// out(r, c) (+)= sum over slices, in slice order (deterministic)
__global__ void splitk_reduce_kernel(GemmArgs g, int nslices) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.M * g.N) return;
    if (g.guard && *g.guard != g.run_if) return;
    const int64_t row = idx / g.N, col = idx - row * g.N;
    double sacc = 0.0;
    for (int z = 0; z < nslices; ++z) sacc += g.part[(int64_t)z * g.M * g.N + idx];
    double *dst = (col < g.ncols0) ? g.out0 + row * g.ld0 + col : g.out1 + row * g.ld1 + (col - g.ncols0);
    *dst = g.accumulate ? *dst + sacc : sacc;
}

// part_ws / part_cap: split-K scratch (doubles); a product whose tile grid would not fill the chip ~4x over
// is cut along K into enough slices to do so.
// out[r][c] += part[0][r][c] + part[1][r][c] + ... (the further K-slices of a split int8 contraction, in slice order)
__global__ void add_partial_kernel(double *out, int64_t ldo, const double *part, int64_t M, int64_t N, int nparts, const int *guard) {
    if (guard && *guard != 0) return;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int64_t r = idx / N, c = idx - r * N;
    double v = out[r * ldo + c];
    for (int z = 0; z < nparts; ++z) v += part[(int64_t)z * M * N + idx];
    out[r * ldo + c] = v;
}

template <int TM, bool BUF>
static hipError_t launch_i8gemm_t(const I8GemmArgs &g, dim3 grid, hipStream_t s) {
    constexpr int nbuf = (BUF && I8_RING != 0) ? (TM == 256 ? 4 : 3) : 2;
    const size_t lds = (size_t)nbuf * (TM + GQS * 32) * 64 + (nbuf >= 3 ? 64 : 0);  // (+ the ring's counters)
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (lds > 65536 && !(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&i8gemm_kernel<64, TM, BUF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((i8gemm_kernel<64, TM, BUF>), grid, dim3(2 * TM), lds, s, g);
    return hipGetLastError();
}

// The 256-row tile also for the statistics product when it has >= 1024 rows (d >= 1024): its B operand -- the digit planes of wP,
// 8 bytes per entry like the fp64 values they stand for -- is re-read once per row block, so half as many row blocks halve the
// dominant operand traffic (config 4: 234.8 -> 222.6 ms).  PPCA_I8GEMM_S256=0: off.
static bool i8gemm_s256() {
    static const bool v = [] {
        const char *e = getenv("PPCA_I8GEMM_S256");
        return !(e && atoi(e) == 0);
    }();
    return v;
}
static hipError_t launch_i8gemm(const I8GemmArgs &g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    static const int tm = [] {  // PPCA_I8GEMM_TM=128: the 4-wave tile everywhere (A/B runs)
        const char *e = getenv("PPCA_I8GEMM_TM");
        return (e && atoi(e) == 128) ? 128 : 256;
    }();
    static const bool nobuf = [] {  // PPCA_I8GEMM_PTR=1: operands by pointer arithmetic (A/B runs; the form for >= 2 GiB)
        const char *e = getenv("PPCA_I8GEMM_PTR");
        return e && atoi(e) == 1;
    }();
    // K is walked in 64-byte steps; both callers pad their rows to that
    const bool buf = !nobuf && g.K % 64 == 0 && (g.ksplit % 64) == 0 && g.M * g.lda < (int64_t(1) << 31) &&
                     GQS * g.plane < (int64_t(1) << 31);
    hipError_t e;
    const bool s256 = i8gemm_s256();
    // XCD-aware tile order (round 5; PPCA_I8GEMM_XCD=0: the plain 2-D order of rounds 1-4).  Measured at config 4's shape
    // (profiles/r05/traffic_cfg4*.json): in the plain order the statistics product fetched its B stripes -- the digit planes of
    // wP, 1.45 GB per chunk -- once per ROW BLOCK, because the four row blocks of a column block (linear ids x, x + 65, x + 130,
    // x + 195) land on four different XCDs, i.e. four L2s: 190 KB of fabric traffic per sample and EM step in this kernel, 65 KB
    // in the XCD-aware order (the whole pipeline: 304 -> 179 KB per sample).  The launch itself gains only 5 % (1.94 -> 1.84 ms: its
    // K loop is not bound by that traffic); with fewer than two column blocks per XCD the order would idle XCDs
    // (d = 512, k = 10, two column blocks: 12.4 against 8.8 ms per iteration), so it needs >= 16.
    static const bool xcd = [] {
        const char *e = getenv("PPCA_I8GEMM_XCD");
        return !(e && atoi(e) == 0);
    }();
    I8GemmArgs h = g;
    const bool tall = tm == 256 && g.tile_rows != 128 && ((g.ksplit == 0 && g.M >= 4096) || (s256 && g.M >= 1024));  // (the Gram: one row per sample; S at d >= 1024)
    const int tmr = tall ? 256 : 128;
    h.ncb = (int)((g.N + 31) / 32);
    h.nrb = (int)((g.M + tmr - 1) / tmr);
    const int nz = g.ksplit > 0 ? g.nsplit : 1;
    h.xcd_map = (xcd && h.ncb >= 16) ? 1 : 0;
    dim3 grid((unsigned)h.ncb, (unsigned)h.nrb, (unsigned)nz);
    if (h.xcd_map) grid = dim3((unsigned)(8 * ((h.ncb + 7) / 8) * h.nrb * nz), 1u, 1u);
    if (tall) e = buf ? launch_i8gemm_t<256, true>(h, grid, s) : launch_i8gemm_t<256, false>(h, grid, s);
    else e = buf ? launch_i8gemm_t<128, true>(h, grid, s) : launch_i8gemm_t<128, false>(h, grid, s);
    if (e != hipSuccess) return e;
    if (g.ksplit > 0) {
        const int64_t tot = g.M * g.N;
        hipLaunchKernelGGL(add_partial_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, g.out, g.ldo, g.out2, g.M, g.N,
                           g.nsplit - 1, g.guard);
    }
    return hipGetLastError();
}

// PPCA_GENERIC_SKINNY=0: the two skinny statistics products through gemm_kernel<2> / <3> at every k (A/B runs)
static bool skinny_ok() {
    static const bool v = [] {
        const char *e = getenv("PPCA_GENERIC_SKINNY");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

// PPCA_GENERIC_FP64=1: both large contractions on the fp64 MFMA always (A/B runs)
static bool generic_i8() {
    static const bool v = [] {
        const char *e = getenv("PPCA_GENERIC_FP64");
        return !(e && atoi(e) == 1);
    }();
    return v;
}

template <int AMODE>
static hipError_t launch_gemm(GemmArgs g, hipStream_t s, int n_cu = 256, double *part_ws = nullptr,
                              int64_t part_cap = 0) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    // (a 128 x 128 tile variant -- 2 x 2 waves of 64 x 64 -- measured slower: 11.3 vs 8.7 ms for the Gram product,
    //  and with K = samples it leaves too few tiles for the chip)
    constexpr int T = 64;
    const int64_t tiles = ((g.N + T - 1) / T) * ((g.M + T - 1) / T);
    int64_t ksplit = 1;
    if (part_ws && tiles < 4 * (int64_t)n_cu) {
        ksplit = (8 * (int64_t)n_cu + tiles - 1) / tiles;
        ksplit = std::min<int64_t>(ksplit, std::max<int64_t>(1, g.K / 256));
        ksplit = std::min<int64_t>(ksplit, part_cap / std::max<int64_t>(1, g.M * g.N));
        if (ksplit > 64) ksplit = 64;
    }
    if (ksplit <= 1) {
        g.part = nullptr;
        dim3 grid((unsigned)((g.N + T - 1) / T), (unsigned)((g.M + T - 1) / T));
        hipLaunchKernelGGL((gemm_kernel<AMODE>), grid, dim3(256), 0, s, g);
        return hipGetLastError();
    }
    g.part = part_ws;
    g.kslice = ((g.K + ksplit - 1) / ksplit + 15) / 16 * 16;
    const int nsl = (int)((g.K + g.kslice - 1) / g.kslice);
    dim3 grid((unsigned)((g.N + T - 1) / T), (unsigned)((g.M + T - 1) / T), (unsigned)nsl);
    hipLaunchKernelGGL((gemm_kernel<AMODE>), grid, dim3(256), 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t tot = g.M * g.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, g, nsl);
    return hipGetLastError();
}

static size_t solve_lds(int k) { return sizeof(double) * 2 * (size_t)(2 * k * (k | 1) + 64); }

static hipError_t set_solve_lds(int k) {
    static size_t cur = 0;
    const size_t need = solve_lds(k);
    if (need > cur) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
        if (e != hipSuccess) return e;
        cur = need;
    }
    return hipSuccess;
}

#define GTRY(expr)                        \
    do {                                  \
        hipError_t _e = (expr);           \
        if (_e != hipSuccess) return _e;  \
    } while (0)

// E-step + statistics of all rows into stats (overwritten).  post == true: only the solve outputs.
// PPCA_GENERIC_PREP=0: rowstats_kernel, gen_maskbytes_kernel and gemm_kernel<1> as three passes over X (A/B runs)
static bool prep_enabled() {
    static const bool v = [] {
        const char *e = getenv("PPCA_GENERIC_PREP");
        return !(e && atoi(e) == 0);
    }();
    return v;
}
// PPCA_EM16=0: the split pipeline below also for 11 <= k <= 16, d <= 256 (A/B runs against ppca_em16.hip)
static bool em16_enabled() {
    static const bool v = [] {
        const char *e = getenv("PPCA_EM16");
        return !(e && atoi(e) == 0);
    }();
    return v;
}
__global__ void flag_any_kernel(const int *flags, int nflags, int *out) {
    int any = 0;
    for (int i = 0; i < nflags; ++i) any |= flags[i];
    *out = any ? 1 : 0;
}

// The EM pass for 11 <= k <= 16, d <= 256: per chunk, the fused E-step kernel and the mask-side statistics kernel of
// ppca_em16.hip (int8-sliced Gram behind the dynamic-range guard; a model that trips it gets its Gram rows from the
// fp64 product below, a guarded launch that otherwise returns at once), partial statistics summed into `stats`.
static hipError_t run_em16(const double *X, int64_t ldx, const double *w, int64_t n, int d, int k, const double *model,
                           double *stats, const GenWs &W, int n_cu, hipStream_t s) {
    const int64_t kp = (int64_t)k * (k + 1) / 2;
    const StatsLayout L(d, k);
    const int ntp = (int)((kp + 15) / 16);
    double *qscale = W.e16_q;
    int *qflag = reinterpret_cast<int *>(qscale + 160);
    int *anyflag = qflag + 16;
    signed char *qtab = reinterpret_cast<signed char *>(qscale + 192);
    GTRY(launch_qprep16(k, model, d, qscale, qtab, qflag, s));
    hipLaunchKernelGGL(flag_any_kernel, dim3(1), dim3(1), 0, s, qflag, ntp, anyflag);
    {
        const int64_t tot = (int64_t)d * kp;
        hipLaunchKernelGGL(qtab_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, model, d, k, W.Q);
        GTRY(hipGetLastError());
    }
    for (int64_t r0 = 0; r0 < n; r0 += W.chunk) {
        const int64_t nc = std::min(W.chunk, n - r0);
        const double *Xc = X + r0 * ldx;
        GemmArgs g{};
        g.X = Xc; g.ldx = ldx; g.mean = model + MODEL_HDR + (int64_t)d * k;
        g.B = W.Q; g.ldb = kp; g.M = nc; g.N = kp; g.K = d;
        g.out0 = W.G; g.ld0 = kp; g.ncols0 = kp; g.out1 = nullptr; g.ld1 = 0; g.accumulate = 0;
        g.guard = anyflag; g.run_if = 1;
        GTRY(launch_gemm<0>(g, s, n_cu, nullptr, 0));
        int grid = fused_grid(nc, n_cu);
        if (grid > E16_MAX_GRID) grid = E16_MAX_GRID;
        Em16Launch a{};
        a.X = Xc; a.ldx = ldx; a.w = w ? w + r0 : nullptr; a.n = nc; a.d = d; a.model = model;
        a.part = W.e16_part; a.qtab = qtab; a.qscale = qscale; a.qflag = qflag; a.Gext = W.G;
        a.Wrows = W.e16_W; a.Mb = W.e16_Mb; a.no_llk = 0; a.dbg = nullptr;
#ifdef PPCA_PHASE_TIMING
        double *dbg = nullptr;
        GTRY(hipMalloc(&dbg, sizeof(double) * 16 * (size_t)grid));
        GTRY(hipMemsetAsync(dbg, 0, sizeof(double) * 16 * (size_t)grid, s));
        a.dbg = dbg;
#endif
        GTRY(launch_em16(k, grid, a, s));
        GTRY(launch_reduce_partials(W.e16_part, grid, L.len, stats, s, r0 > 0 ? 1 : 0));
#ifdef PPCA_PHASE_TIMING
        {
            std::vector<double> h((size_t)grid * 16);
            GTRY(hipMemcpyAsync(h.data(), dbg, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s));
            GTRY(hipStreamSynchronize(s));
            GTRY(hipFree(dbg));
            double t[16] = {0};
            for (int g2 = 0; g2 < grid; ++g2)
                for (int i = 0; i < 16; ++i) t[i] += h[(size_t)g2 * 16 + i] / grid;
            const double tiles = (double)((nc + 31) / 32) / grid;
            fprintf(stderr, "[em16 estep cycles/tile] P2: b %.0f  Gram %.0f  stores+barrier %.0f | P3: load+barrier %.0f  factor %.0f  factor to LDS+barrier %.0f  substitutions %.0f  scalars+barrier %.0f | P4a: cross + rows to HBM %.0f  barrier %.0f | staging (+ C fragments) + barrier %.0f  (tiles/WG %.1f; k <= 13: load .. substitutions are one figure)\n",
                    t[0] / tiles, t[1] / tiles, t[2] / tiles, t[3] / tiles, t[4] / tiles, t[5] / tiles, t[6] / tiles, t[7] / tiles, t[8] / tiles, t[9] / tiles, t[10] / tiles, tiles);
        }
#endif
    }
    return hipSuccess;
}

static hipError_t generic_run(const double *X, int64_t ldx, const double *w, int64_t n, int d, int k,
                              const double *model, bool em, double *stats,
                              double *scal8, double *llks, double *states, double *covs, double *recon, int recon_mode,
                              void *ws, int n_cu, hipStream_t s) {
    const int64_t kp = (int64_t)k * (k + 1) / 2;
    const StatsLayout L(d, k);
    GenWs W = carve(ws, d, k, n);
    const double *mean = model + MODEL_HDR + (int64_t)d * k;
    const double *Cm = model + MODEL_HDR;
    // (round 6) k = 65 .. 128 run the int8-sliced contractions too (their tables, guards and the int8 GEMM are not bound in k; only the
    // fused pre-solve pass is: those sizes take the three separate passes) -- PPCA_GENERIC_BIG_FP64=1: the fp64 contractions of round 5
    static const bool big_fp64 = [] {
        const char *e = getenv("PPCA_GENERIC_BIG_FP64");
        return e && atoi(e) == 1;
    }();
    const bool i8 = generic_i8() && (k <= 64 || !big_fp64);
    if (em && i8 && em16_enabled() && em16_covers(d, k)) return run_em16(X, ldx, w, n, d, k, model, stats, W, n_cu, s);
    {
        const int64_t tot = (int64_t)d * kp;
        hipLaunchKernelGGL(qtab_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, model, d, k, W.Q);
        GTRY(hipGetLastError());
        if (i8) {  // digit planes of Q and the Gram guard flag (flags[0]) of this model
            hipLaunchKernelGGL(gen_rmin_kernel, dim3(1), dim3(256), 0, s, model, d, k, W.rmin, W.flags);
            hipLaunchKernelGGL(gen_qdigits_kernel, dim3((unsigned)kp), dim3(256), 0, s, model, d, k, W.dpad, (int)kp, W.rmin,
                               W.scaleQ, W.BtQ, W.flags);
            GTRY(hipGetLastError());
        }
    }
    if (k <= 64) GTRY(set_solve_lds(k));
    if (em) GTRY(hipMemsetAsync(stats, 0, sizeof(double) * (size_t)L.len, s));
    for (int64_t r0 = 0; r0 < n; r0 += W.chunk) {
        const int64_t nc = std::min(W.chunk, n - r0);
        const double *Xc = X + r0 * ldx;
        const double *wc = w ? w + r0 : nullptr;
        GemmArgs g{};
        g.X = Xc; g.ldx = ldx; g.mean = mean;
        const int64_t ncpad = pad64(nc);
        const bool prep = prep_enabled() && k <= 64;
        if (prep) {  // row statistics, mask bytes and b = X~ C in one pass over the chunk's rows
            const dim3 pg((unsigned)(ncpad / 64));
            if (k <= 16) hipLaunchKernelGGL((gen_prep_kernel<1>), pg, dim3(256), prep_lds(1), s, Xc, ldx, nc, d, W.dpad, W.npad, model, k, W.A, W.AT, W.xx, W.mc, W.Bz, i8 ? 1 : 0);
            else if (k <= 32) hipLaunchKernelGGL((gen_prep_kernel<2>), pg, dim3(256), prep_lds(2), s, Xc, ldx, nc, d, W.dpad, W.npad, model, k, W.A, W.AT, W.xx, W.mc, W.Bz, i8 ? 1 : 0);
            else {
                static std::atomic<unsigned long long> prep_done{0ull};  // (per device, as the other kernels above 64 KB of LDS)
                int dev = 0;
                GTRY(hipGetDevice(&dev));
                const unsigned long long bit = 1ull << (dev & 63);
                if (!(prep_done.load(std::memory_order_acquire) & bit)) {
                    GTRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&gen_prep_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)prep_lds(4)));
                    prep_done.fetch_or(bit, std::memory_order_release);
                }
                hipLaunchKernelGGL((gen_prep_kernel<4>), pg, dim3(256), prep_lds(4), s, Xc, ldx, nc, d, W.dpad, W.npad, model, k, W.A, W.AT, W.xx, W.mc, W.Bz, i8 ? 1 : 0);
            }
            GTRY(hipGetLastError());
        } else {
            hipLaunchKernelGGL(rowstats_kernel, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, s, Xc, ldx, nc, d, model, k,
                               W.xx, W.mc);
            GTRY(hipGetLastError());
        }
        if (i8) {
            if (!prep) {
                dim3 mg((unsigned)(W.dpad / 64), (unsigned)(ncpad / 64));
                hipLaunchKernelGGL(gen_maskbytes_kernel, mg, dim3(256), 0, s, Xc, ldx, nc, d, W.dpad, W.npad, W.A, W.AT);
                GTRY(hipGetLastError());
            }
            // G = Mask . Q on the int8 MFMA (exact integer accumulation) unless the guard raised flags[0]
            I8GemmArgs q{};
            q.A = W.A; q.lda = W.dpad; q.Bt = W.BtQ; q.ldb = W.dpad; q.plane = kp * (int64_t)W.dpad;
            static const int gram_tile = [] {  // PPCA_I8GEMM_GRAM_TM=128: the Gram product on 128-row tiles (A/B runs)
                const char *e = getenv("PPCA_I8GEMM_GRAM_TM");
                return (e && atoi(e) == 128) ? 128 : 0;
            }();
            q.tile_rows = gram_tile;
            q.M = nc; q.N = kp; q.K = W.dpad; q.scale = W.scaleQ; q.out = W.G; q.ldo = kp; q.accumulate = 0;
            q.guard = W.flags;
            GTRY(launch_i8gemm(q, s));
        }
        // G = Mask . Q (fp64 MFMA; with the int8 form enabled: only when its guard tripped)
        g.B = W.Q; g.ldb = kp; g.M = nc; g.N = kp; g.K = d;
        g.out0 = W.G; g.ld0 = kp; g.ncols0 = kp; g.out1 = nullptr; g.ld1 = 0; g.accumulate = 0;
        g.guard = i8 ? W.flags : nullptr; g.run_if = 1;
        GTRY(launch_gemm<0>(g, s));
        g.guard = nullptr;
        if (!prep) {  // b = X~ . C
            g.B = Cm; g.ldb = k; g.N = k;
            g.out0 = W.Bz; g.ld0 = k + 1; g.ncols0 = k;
            GTRY(launch_gemm<1>(g, s));
        }
        SolveArgs a{};
        a.G = W.G; a.Bz = W.Bz; a.xx = W.xx; a.mc = W.mc; a.w = wc; a.n = nc; a.k = k;
        a.model = model; a.sc = W.sc; a.em = em ? 1 : 0;
        a.llks = llks ? llks + r0 : nullptr;
        a.states = states ? states + r0 * k : nullptr;
        a.covs = covs ? covs + r0 * (int64_t)k * k : nullptr;
        a.need_sigma = (covs || (recon && recon_mode >= 2)) ? 1 : 0;
        if (getenv("PPCA_GENERIC_LDS_SOLVE")) {  // the LDS-resident variant, kept for A/B runs
            int sgrid = (int)std::min<int64_t>((nc + 1) / 2, (int64_t)n_cu);
            if (sgrid < 1) sgrid = 1;
            hipLaunchKernelGGL(solve_kernel, dim3(sgrid), dim3(128), solve_lds(k), s, a);
            GTRY(hipGetLastError());
        } else {
            GTRY(launch_solve(a, n_cu, s));
        }
        double *scal = em ? stats + L.scalars : scal8;
        {
            const int sb = (int)((nc + SCAL_ROWS - 1) / SCAL_ROWS);
            hipLaunchKernelGGL(scal_reduce_kernel, dim3((unsigned)sb), dim3(256), 0, s, W.sc, wc, nc, W.spart);
            hipLaunchKernelGGL(scal_final_kernel, dim3(1), dim3(256), 0, s, W.spart, sb, scal, (em || r0 > 0) ? 1 : 0);
            GTRY(hipGetLastError());
        }
        if (em) {
            if (i8) {
                // column scales of wP over the chunk + its guard (flags[1]), digit planes, S += Mask^T . wP on the int8 MFMA
                // (round 6) from the call's second chunk on the digits are cut in the same pass that takes the column statistics, under
                // the scales predicted from the chunk before (gen_wdigits_pred_kernel); the first chunk of a call takes the statistics
                // first -- so that a result never depends on an earlier call.  PPCA_GEN_WPRED=0: every chunk as the first.
                static const bool wpred = [] {
                    const char *e = getenv("PPCA_GEN_WPRED");
                    return !(e && atoi(e) == 0);
                }();
                const int nb = (int)((nc + 255) / 256);
                dim3 dg((unsigned)((kp + 63) / 64), (unsigned)(ncpad / 64));
                const dim3 lg((unsigned)((kp + 31) / 32), (unsigned)nb);
                if (wpred && r0 > 0) {
                    hipLaunchKernelGGL(gen_wdigits_lines_kernel, lg, dim3(256), 0, s, W.G, nc, ncpad, (int)kp, W.npad, W.predW,
                                       W.BtW, W.colpart, W.flags, 1);
                    hipLaunchKernelGGL(gen_colscale_kernel, dim3((unsigned)kp), dim3(256), 0, s, W.colpart, nb, nc, (int)kp, W.scaleW,
                                       W.flags, W.predW, W.redoW, 1);
                    hipLaunchKernelGGL(gen_wdigits_kernel, dim3(dg.x, std::min<unsigned>(dg.y, 256u)), dim3(256), 0, s, W.G, nc, ncpad, (int)kp,
                                       W.npad, W.scaleW, W.BtW, W.flags, W.redoW);
                } else {
                    hipLaunchKernelGGL(gen_colstat_kernel, dim3((unsigned)((kp + 255) / 256), (unsigned)nb), dim3(256), 0, s, W.G,
                                       nc, (int)kp, W.colpart, W.flags);
                    hipLaunchKernelGGL(gen_colscale_kernel, dim3((unsigned)kp), dim3(256), 0, s, W.colpart, nb, nc, (int)kp, W.scaleW,
                                       W.flags, W.predW, (int *)nullptr, 0);
                    if (wpred)
                        hipLaunchKernelGGL(gen_wdigits_lines_kernel, lg, dim3(256), 0, s, W.G, nc, ncpad, (int)kp, W.npad,
                                           W.scaleW, W.BtW, W.colpart, W.flags, 0);
                    else
                        hipLaunchKernelGGL(gen_wdigits_kernel, dg, dim3(256), 0, s, W.G, nc, ncpad, (int)kp, W.npad, W.scaleW, W.BtW,
                                           W.flags, (const int *)nullptr);
                }
                GTRY(hipGetLastError());
                I8GemmArgs q{};
                q.A = W.AT; q.lda = W.npad; q.Bt = W.BtW; q.ldb = W.npad; q.plane = kp * W.npad;
                q.M = d; q.N = kp; q.K = ncpad; q.scale = W.scaleW; q.out = stats + L.S; q.ldo = kp; q.accumulate = 1;
                q.guard = W.flags + 1;
                bool launched = false;
                {   // a grid a little above the tiles the chip runs at once spends a nearly empty extra round
                    // (two 128-row workgroups fit a CU, one 256-row workgroup: launch_i8gemm picks the tile as below)
                    const bool t256 = q.M >= 1024 && i8gemm_s256();
                    const int64_t tmr = t256 ? 256 : 128, nrb = (q.M + tmr - 1) / tmr, ncb = (q.N + 31) / 32;
                    const int64_t tiles = nrb * ncb, slots = (t256 ? 1 : 2) * (int64_t)n_cu;
                    const int64_t cb_round = slots / nrb;                       // column blocks one full round of the chip takes
                    const int64_t rest = cb_round > 0 ? ncb % cb_round : 0;     // ... and those left for a last, partly empty one
                    if (tiles > slots && rest > 0 && rest * nrb * 4 <= slots && ncpad >= 8192) {
                        // (round 5) config 4: 65 column blocks x 4 row blocks = 260 tiles on 256 CUs.  Rounds 3-4 cut K in two (520
                        // workgroups: three rounds of half a tile, 1.5 tile times); now the 64 column blocks that fill the chip
                        // exactly run whole, and the last one is cut along K into as many slices as fill it once more (64): 1.02
                        // tile times.  Both launches go through the same kernel; the slices are added in slice order.
                        I8GemmArgs q1 = q;
                        q1.N = (ncb - rest) * 32;
                        GTRY(launch_i8gemm(q1, s));
                        I8GemmArgs q2 = q;
                        const int64_t c0 = q1.N;
                        q2.Bt = q.Bt + c0 * q.ldb;
                        q2.scale = q.scale + c0;
                        q2.out = q.out + c0;
                        q2.N = q.N - c0;
                        int64_t ns = slots / (rest * nrb);
                        ns = std::min<int64_t>(ns, ncpad / 1024);                                         // >= 1024 samples per slice
                        ns = std::min<int64_t>(ns, W.part_cap / std::max<int64_t>(1, q2.M * q2.N) + 1);  // partials that fit
                        if (ns > 64) ns = 64;
                        if (ns >= 2) {
                            q2.ksplit = ((ncpad + ns - 1) / ns + 63) / 64 * 64;
                            q2.nsplit = (int)((ncpad + q2.ksplit - 1) / q2.ksplit);
                            q2.out2 = W.part;
                        }
                        GTRY(launch_i8gemm(q2, s));
                        launched = true;
                    } else if (tiles < slots) {
                        // ... and a grid far below it leaves the chip idle: cut the samples into enough slices to fill it
                        int64_t ns = (slots + tiles - 1) / tiles;
                        ns = std::min<int64_t>(ns, ncpad / 4096);                                   // >= 4096 samples per slice
                        ns = std::min<int64_t>(ns, W.part_cap / std::max<int64_t>(1, q.M * q.N) + 1);  // partials that fit
                        if (ns > 64) ns = 64;
                        if (ns >= 2) {
                            q.ksplit = ((ncpad + ns - 1) / ns + 63) / 64 * 64;
                            q.nsplit = (int)((ncpad + q.ksplit - 1) / q.ksplit);
                            q.out2 = W.part;
                        }
                    }
                }
                if (!launched)                 GTRY(launch_i8gemm(q, s));
            }
            // S += Mask^T . wP (fp64 MFMA; with the int8 form enabled: only when the chunk's guard tripped)
            g.B = W.G; g.ldb = kp; g.M = d; g.N = kp; g.K = nc;
            g.out0 = stats + L.S; g.ld0 = kp; g.ncols0 = kp; g.accumulate = 1;
            g.guard = i8 ? W.flags + 1 : nullptr; g.run_if = 1;
            GTRY(launch_gemm<2>(g, s, n_cu, W.part, W.part_cap));
            g.guard = nullptr;
            hipError_t serr = hipSuccess;
            if (skinny_ok() && launch_skinny_xt(Xc, ldx, nc, d, k, mean, W.Bz, stats, L, W.part, W.part_cap, n_cu, s, &serr)) {
                GTRY(serr);  // both skinny products in one pass over the chunk's rows (k + 1 <= 32 columns)
            } else {
                // [U | totals] += Mask^T . [wz | w]
                g.B = W.Bz; g.ldb = k + 1; g.N = k + 1;
                g.out0 = stats + L.U; g.ld0 = k; g.ncols0 = k; g.out1 = stats + L.totals; g.ld1 = 1;
                GTRY(launch_gemm<2>(g, s, n_cu, W.part, W.part_cap));
                // [cross | sumx] += X~^T . [wz | w]
                g.out0 = stats + L.cross; g.out1 = stats + L.sumx;
                GTRY(launch_gemm<3>(g, s, n_cu, W.part, W.part_cap));
            }
        } else if (recon) {
            static const bool naive = [] {  // PPCA_GENERIC_RECON=naive: one thread per output element (A/B runs)
                const char *e = getenv("PPCA_GENERIC_RECON");
                return e && e[0] == 'n';
            }();
            if (naive || k > 64) {
                const int64_t tot = nc * d;
                hipLaunchKernelGGL(recon_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Xc, ldx, nc, d, k,
                                   model, W.Bz, W.G, recon_mode, recon + r0 * d);
            } else {
                // rows per block: enough blocks to fill the chip a few times over, at least 16 rows to amortise the row of C
                int rpb = (int)std::max<int64_t>(16, std::min<int64_t>(256, nc / (8 * (int64_t)n_cu) + 1));
                const dim3 rg((unsigned)((nc + rpb - 1) / rpb), (unsigned)((d + 255) / 256));
                if (k <= 16) hipLaunchKernelGGL((recon2_kernel<16>), rg, dim3(256), 0, s, Xc, ldx, nc, d, k, model, W.Bz, W.G, recon_mode, recon + r0 * d, rpb);
                else if (k <= 32) hipLaunchKernelGGL((recon2_kernel<32>), rg, dim3(256), 0, s, Xc, ldx, nc, d, k, model, W.Bz, W.G, recon_mode, recon + r0 * d, rpb);
                else hipLaunchKernelGGL((recon2_kernel<64>), rg, dim3(256), 0, s, Xc, ldx, nc, d, k, model, W.Bz, W.G, recon_mode, recon + r0 * d, rpb);
            }
            GTRY(hipGetLastError());
        }
    }
    if (em && i8) {
        static const bool wstats = [] {
            const char *e = getenv("PPCA_GEN_WPRED_STATS");
            return e && atoi(e) == 1;
        }();
        if (wstats) {  // (diagnostic: columns cut a second time since the last report)
            int f = 0;
            GTRY(hipMemcpyAsync(&f, W.flags + 3, sizeof(int), hipMemcpyDeviceToHost, s));
            GTRY(hipMemsetAsync(W.flags + 3, 0, sizeof(int), s));
            GTRY(hipStreamSynchronize(s));
            fprintf(stderr, "[generic wP digits] %d of %lld (column, chunk) pairs cut a second time (chunks of %lld rows)\n", f,
                    (long long)(kp * std::max<int64_t>(0, (n + W.chunk - 1) / W.chunk - 1)), (long long)W.chunk);
        }
    }
    return hipSuccess;
}

// The Gram engine the generic pipeline would use for this model: *forced = 1 when the environment pins the fp64 form,
// otherwise -1 and *flag_dev points at the guard flag (0 = int8-sliced, non-zero = fp64) the digit kernels just wrote.
// ws: a workspace of generic_workspace_bytes(d, k, 1).
hipError_t generic_gram_guard(int d, int k, const double *model, void *ws, hipStream_t s, const int **flag_dev, int *forced) {
    if (!generic_i8() || (k > 64 && getenv("PPCA_GENERIC_BIG_FP64") && atoi(getenv("PPCA_GENERIC_BIG_FP64")) == 1)) {
        *forced = 1;
        return hipSuccess;
    }
    *forced = -1;
    const int64_t kp = (int64_t)k * (k + 1) / 2;
    GenWs W = carve(ws, d, k, 1);
    if (em16_enabled() && em16_covers(d, k)) {  // the EM pass of these shapes: the guard of its own slice table (ppca_em16.hip)
        double *qscale = W.e16_q;
        int *qflag = reinterpret_cast<int *>(qscale + 160);
        GTRY(launch_qprep16(k, model, d, qscale, reinterpret_cast<signed char *>(qscale + 192), qflag, s));
        hipLaunchKernelGGL(flag_any_kernel, dim3(1), dim3(1), 0, s, qflag, (int)((kp + 15) / 16), qflag + 16);
        *flag_dev = qflag + 16;
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gen_rmin_kernel, dim3(1), dim3(256), 0, s, model, d, k, W.rmin, W.flags);
    hipLaunchKernelGGL(gen_qdigits_kernel, dim3((unsigned)kp), dim3(256), 0, s, model, d, k, W.dpad, (int)kp, W.rmin, W.scaleQ, W.BtQ,
                       W.flags);
    *flag_dev = W.flags;
    return hipGetLastError();
}

hipError_t generic_em_accumulate(const double *X, int64_t ldx, const double *w, int64_t n, int d, int k,
                                 const double *model, double *stats, void *ws, int n_cu, hipStream_t s) {
    return generic_run(X, ldx, w, n, d, k, model, true, stats, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                       ws, n_cu, s);
}

hipError_t generic_post(const double *X, int64_t ldx, const double *w, int64_t n, int d, int k, const double *model,
                        double *scal8, double *llks, double *states, double *covs,
                        double *recon, int recon_mode, void *ws, int n_cu, hipStream_t s) {
    if (n == 0) return hipMemsetAsync(scal8, 0, sizeof(double) * 8, s);
    return generic_run(X, ldx, w, n, d, k, model, false, nullptr, scal8, llks, states, covs, recon,
                       recon_mode, ws, n_cu, s);
}

hipError_t generic_finalize(int k, int d, const double *stats, const double *model_in, double *model_out, double tau,
                            int has_ig, double alpha, double beta, int n_cu, hipStream_t s) {
    if (k > 64) {  // one workgroup per dimension (solve_big_kernel's tools)
        const size_t lds = solve_big_lds(k);
        GTRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&gen_rowsolve_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(gen_rowsolve_big_kernel, dim3((unsigned)std::min(d, 4 * n_cu)), dim3(256), lds, s, stats, model_in, model_out, d, k, tau);
        GTRY(hipGetLastError());
        hipLaunchKernelGGL(gen_finalize_misc_kernel, dim3(1), dim3(256), 0, s, stats, model_in, model_out, d, k, has_ig, alpha, beta);
        return hipGetLastError();
    }
    const size_t lds = sizeof(double) * 2 * (size_t)(k * (k | 1) + 64);
    static size_t cur = 0;
    if (lds > cur) {
        GTRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&gen_rowsolve_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cur = lds;
    }
    int grid = std::min((d + 1) / 2, n_cu * 2);
    hipLaunchKernelGGL(gen_rowsolve_kernel, dim3(grid), dim3(128), lds, s, stats, model_in, model_out, d, k, tau);
    GTRY(hipGetLastError());
    hipLaunchKernelGGL(gen_finalize_misc_kernel, dim3(1), dim3(256), 0, s, stats, model_in, model_out, d, k, has_ig,
                       alpha, beta);
    return hipGetLastError();
}

}  // namespace ppca
