// ppca_solve4.hip -- per-sample solve of the split pipeline for 17 <= k <= 64, several samples per wave.
//
// Same stage and the same mathematics as solve_mfma_body (ppca_generic.hip): M = G + sigma^2 I (reference:
// ppca/src/ppca_model.rs:195-208 infer_one: the masked posterior's k x k system; :142-149 llk), padded with an identity
// block to n = 16 NB, inverted IN PLACE in LDS by the three blocked sweeps over 16 x 16 blocks
//   potrf   for j: T_jj = chol(A_jj)^-1 (kept in the diagonal block); L_ij = A_ij T_jj^T (i > j); A_il -= L_ij L_lj^T
//   trtri   W = L^-1: for j descending, i descending: W_ij = -(sum_{t=j+1..i} W_it L_tj) T_jj
//   lauum   M^-1 = W^T W
// with every block product on v_mfma_f64_16x16x4.  What changes is the mapping.  solve_mfma_body gives a wave ONE sample:
// the 16 x 16 diagonal blocks are factored by 16 lanes (the other 48 repeat them), every pivot step goes through an LDS
// write and its read-back, and each block product waits for its own operands and its own four dependent MFMAs -- a chain of
// latencies, 43 k cycles per sample and wave at n = 32 (profiles/r04: 17 ms of a 28 ms EM iteration at N = 2 M, k = 20).
// Here a wave owns NS samples (4 at n = 32, 2 at n = 48 / 64: what LDS holds at one wave per SIMD):
//   * diagonal blocks: lane = (sample lane / 16, row lane % 16) -- the four 16-lane rows of the wave factor four samples at
//     once, the pivot column and the rows of T travel by DPP row_newbcast (a broadcast inside each 16-lane row: no LDS,
//     no scalar round trip), L and T stay in registers, the block leaves as T with a zero upper part;
//   * block products: the NS samples' products of one step are issued together -- 8 NS operand reads in flight, then 4 NS
//     MFMAs of which consecutive ones are independent;
//   * z = M^-1 b with lane = (sample, row), the traces by 16-lane row reductions (DPP row_ror);
//   * the packed index -> (LDS offset, row, column) map is a table built once per workgroup.
// The padding (k < n) is re-initialised for every group of samples (a non-finite entry must not leak into the next group).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ppca_solve.hpp"
#include "ppca_device.hpp"  // lean_log

namespace ppca {
#ifdef S4_TIMING  // tools/s4bench: cycles per phase, summed over the groups of wave 0 of every workgroup; [15] = groups counted
__device__ unsigned long long s4_dbg[16];
#define S4_STAMP(p)                                                      \
    do { /* (sums stay in registers: an atomic per stamp is a vector-memory operation the kernel's explicit vmcnt waits would wait for) */ \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        tacc_[p] += now_ - stamp_;                                       \
        stamp_ = __builtin_amdgcn_s_memtime();                           \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#else
#define S4_STAMP(p)
#endif
namespace {

typedef double d4s_t __attribute__((ext_vector_type(4)));
typedef double d2s_t __attribute__((ext_vector_type(2)));
typedef unsigned u2s_t __attribute__((ext_vector_type(2)));
constexpr int S4_LD = 18, S4_BSZ = 16 * S4_LD;
constexpr int s4_ns(int nb) { return nb <= 2 ? 4 : 2; }       // samples per wave
constexpr int s4_waves(int nb) { return nb == 4 ? 3 : 4; }  // waves per workgroup (one workgroup per CU: what LDS holds at n = 64; a fifth
                                                            // wave at n = 48 halves every wave's register budget: 20.3 against 10.8 ms per million rows)
constexpr int s4_samp(int nb) { return nb * (nb + 1) / 2 * S4_BSZ + 2 * 16 * nb; }  // doubles per sample: blocks | z | b
constexpr int s4_blk(int i, int j) { return (i * (i + 1) / 2 + j) * S4_BSZ; }        // block row i >= block column j

// step n of the lauum sweep: row ii ascending, l <= ii ascending, t = ii .. nb - 1
struct S4Step {
    int ii, l, t;
};
constexpr S4Step s4_lauum_step(int nb, int n) {
    for (int ii = 0; ii < nb; ++ii)
        for (int l = 0; l <= ii; ++l)
            for (int t = ii; t < nb; ++t, --n)
                if (n == 0) return S4Step{ii, l, t};
    return S4Step{0, 0, 0};
}

template <int I, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(f);
    }
}
// lane p of every 16-lane row to all lanes of that row (v_mov_b64_dpp: row_newbcast is the one DPP control 64-bit moves take)
template <int P>
__device__ __forceinline__ double rowbcast(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + P, 0xf, 0xf, true);
}
template <int R>
__device__ __forceinline__ double rowror(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x120 + R, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + R, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row16_sum(double v) {  // sum over the 16 lanes of a row, in every lane of it
    v += rowror<8>(v);
    v += rowror<4>(v);
    v += rowror<2>(v);
    v += rowror<1>(v);
    return v;
}

// One 16 x 16 diagonal block per 16-lane row: D (lower triangle; what lies above the diagonal is never used) -> T = chol(D)^-1,
// lower, zero upper part.  lane l15 = row.  mant / ex collect det(L)^2 (the pivots) as a mantissa product and an exponent sum.
__device__ __forceinline__ void s4_diag_block(double *D, int l15, double &mant, int &ex) {
    constexpr int LD = S4_LD;
    double row[16];
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2) {
        const d2s_t v = *reinterpret_cast<const d2s_t *>(D + l15 * LD + 2 * c2);
        row[2 * c2] = v.x;
        row[2 * c2 + 1] = v.y;
    }
    double rinv_own = 0.0;
    double ls[16];  // L[r][p] / L_pp: the scaled column the triangular inverse below multiplies with
    // right-looking Cholesky, lane = row: after step p, row[p] = L[r][p] (r > p); entries above the diagonal are never read
    sfor<0, 16>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        const double piv = rowbcast<p>(row[p]);
        const double rinv = fast_rsqrt(piv);
        int e;
        mant *= frexp(piv, &e);
        ex += e;
        rinv_own = (l15 == p) ? rinv : rinv_own;
        row[p] *= rinv;
        ls[p] = (p < l15) ? row[p] * rinv : 0.0;  // strict lower part only (the update below then needs no lane mask)
        sfor<p + 1, 16>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            row[c] = fma(-row[p], rowbcast<c>(row[p]), row[c]);
        });
    });
    // T = L^-1 row by row, T[r][c] = V[r][c] / L_rr with V[r][r] = 1 and V[r][c] = -sum_{t = c .. r-1} (L[r][t] / L_tt) V[t][c]:
    // at step t lane t holds its final row of V and hands it to the rows below
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
    sfor<0, 15>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        sfor<0, t>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            v[c] = fma(-ls[t], rowbcast<t>(v[c]), v[c]);
        });
        v[t] = -ls[t];  // (V[t][t] = 1)
    });
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2) {
        d2s_t o;
        const int c0 = 2 * c2, c1 = 2 * c2 + 1;
        o.x = (c0 < l15) ? v[c0] * rinv_own : (c0 == l15 ? rinv_own : 0.0);
        o.y = (c1 < l15) ? v[c1] * rinv_own : (c1 == l15 ? rinv_own : 0.0);
        *reinterpret_cast<d2s_t *>(D + l15 * LD + 2 * c2) = o;
    }
}

template <int NB, bool EM>
__global__ __launch_bounds__(64 * s4_waves(NB)) void solve4_kernel(SolveArgs a) {
    constexpr int N = 16 * NB, LD = S4_LD, BSZ = S4_BSZ, NBK = NB * (NB + 1) / 2, NS = s4_ns(NB), W = s4_waves(NB), SAMP = s4_samp(NB);
    constexpr int ZOFF = NBK * BSZ, BOFF = ZOFF + N;
    constexpr int NPK = (N * (N + 1) / 2 + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    unsigned *tab = reinterpret_cast<unsigned *>(gsm + W * NS * SAMP);
    const int k = a.k, kp = k * (k + 1) / 2;
    // packed index e -> LDS offset of (r, c) | r << 12 | c << 18 | (r == c) << 24
    for (int e = threadIdx.x; e < kp; e += 64 * W) {
        int r = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        while (r * (r + 1) / 2 > e) --r;
        const int c = e - r * (r + 1) / 2;
        const int rb = r >> 4, cb = c >> 4;
        const int off = (rb * (rb + 1) / 2 + cb) * BSZ + (r & 15) * LD + (c & 15);
        tab[e] = (unsigned)off | (unsigned)r << 12 | (unsigned)c << 18 | (r == c ? 1u << 24 : 0u);
    }
    __syncthreads();
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: group index, row pointers and descriptors in SGPRs)
    double *Aw = gsm + wave * NS * SAMP;
    const double s2 = a.model[1], lnsig = a.model[2];
    const int64_t ngroups = (a.n + NS - 1) / NS;
    const int64_t gstride = (int64_t)gridDim.x * W;
    constexpr int QC = 8;  // packed entries per lane that travel together (one uniform branch per chunk)
    // A group's inputs are REQUESTED half a group ahead (after the triangular inverse of the previous one) and sit in
    // registers until its LDS image is built: one wave per SIMD has nobody else to cover a global-memory latency.
    double gv[NS][NPK], bq[NS];
    auto request = [&](int64_t grp) {
        const int64_t i0 = grp * NS;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t is = i0 + s < a.n ? i0 + s : a.n - 1;
            // (buffer descriptor over the sample's packed row: lanes past its end read 0 and need no clamp or mask)
            const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(a.G + is * kp, 0, kp * 8, 0x00020000);
#pragma unroll
            for (int q0 = 0; q0 < NPK; q0 += QC)
                if (64 * q0 < kp) {
#pragma unroll
                    for (int q = q0; q < q0 + QC && q < NPK; ++q) {
                        const u2s_t v = __builtin_amdgcn_raw_buffer_load_b64(gr, lane * 8, q * 512, 0);
                        gv[s][q] = __longlong_as_double(((long long)v.y << 32) | v.x);
                    }
                }
            bq[s] = a.Bz[is * (k + 1) + (lane < k ? lane : k)];
        }
    };
    {
        const int64_t first = (int64_t)blockIdx.x * W + wave;
        if (first < ngroups) request(first);
        __builtin_amdgcn_s_waitcnt(0x0f70);  // (vmcnt(0): see the note at the group's first store -- no load is pending at the top of the loop on either path)
    }
#ifdef S4_TIMING
    unsigned long long tacc_[16] = {0};
#endif
    for (int64_t grp = (int64_t)blockIdx.x * W + wave; grp < ngroups; grp += gstride) {
        asm volatile("" : "+v"(lane));  // (per-lane LDS addresses are recomputed per group instead of being parked across it)
        const int l15 = lane & 15, l4 = lane >> 4;
        const int64_t i0 = grp * NS;
#ifdef S4_TIMING
        unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
        tacc_[15] += 1ull;
#endif
        // ---- the group's LDS image: padding, b, the lower triangle of M = G + s2 I
        if (k < N) {
            for (int e = lane; e < NS * SAMP / 2; e += 64) reinterpret_cast<d2s_t *>(Aw)[e] = d2s_t{0.0, 0.0};
            for (int e = k + lane; e < N; e += 64) {
                const int eb = e >> 4;
#pragma unroll
                for (int s = 0; s < NS; ++s) Aw[s * SAMP + s4_blk(eb, eb) + (e & 15) * (LD + 1)] = 1.0;
            }
        }
        if (lane < N) {
#pragma unroll
            for (int s = 0; s < NS; ++s) Aw[s * SAMP + BOFF + lane] = (lane < k) ? bq[s] : 0.0;
        }
#pragma unroll
        for (int q0 = 0; q0 < NPK; q0 += QC)
            if (64 * q0 < kp) {
                unsigned tv[QC];
#pragma unroll
                for (int j = 0; j < QC && q0 + j < NPK; ++j) {
                    const int e = lane + 64 * (q0 + j);
                    tv[j] = tab[e < kp ? e : kp - 1];
                }
#pragma unroll
                for (int j = 0; j < QC && q0 + j < NPK; ++j) {
                    const int e = lane + 64 * (q0 + j);
                    const unsigned t = tv[j];
                    const int off = e < kp ? (int)(t & 4095) : ZOFF;  // (lanes past the end write the z slot, which is free here)
                    const double sh = (t >> 24) ? s2 : 0.0;
                    // Only the LOWER triangle of a diagonal block is written: its factorisation (lane = row) never lets an entry
                    // above the diagonal reach one below, and the trailing updates change a block element by element.
#pragma unroll
                    for (int s = 0; s < NS; ++s) Aw[s * SAMP + off] = gv[s][q0 + j] + sh;
                }
            }
        S4_STAMP(0);
        // ---- block products of the NS samples.  Operand patterns of v_mfma_f64_16x16x4 (lane = (l15, l4), k-step st):
        //   pattern N: Z[l15][4 st + l4]  = A operand of Z,   or B operand of Z^T
        //   pattern T: Z[4 st + l4][l15]  = A operand of Z^T, or B operand of Z
        // and the accumulator (C layout: row l4 + 4 r, column l15) of a product IS pattern T of the result (register r = step
        // st) -- equivalently pattern N of the result's transpose.  The sweeps below compute a product TRANSPOSED whenever its
        // only reader wants it on the left: L_ij^T feeds the trailing update A_il -= L_ij L_lj^T from registers (both operands),
        // Y^T feeds W_ij = -Y T_jj -- no LDS store / wait / reload between a product and its consumer.
        const int iN = l15 * LD + l4, iT = l4 * LD + l15;
        auto ldN = [&](double(&v)[NS][4], int off) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int st = 0; st < 4; ++st) v[s][st] = Aw[s * SAMP + off + iN + 4 * st];
        };
        auto ldT = [&](double(&v)[NS][4], int off) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int st = 0; st < 4; ++st) v[s][st] = Aw[s * SAMP + off + iT + 4 * st * LD];
        };
        auto fmaN = [&](d4s_t(&acc)[NS], const double(&av)[NS][4], const double(&bv)[NS][4], bool neg) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(neg ? -av[s][st] : av[s][st], bv[s][st], acc[s], 0, 0, 0);
        };
        auto regs = [&](double(&v)[NS][4], const d4s_t(&acc)[NS]) {  // an accumulator as pattern T of its product
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int st = 0; st < 4; ++st) v[s][st] = acc[s][st];
        };
        auto ldC = [&](d4s_t(&acc)[NS], int zo) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s][r] = Aw[s * SAMP + zo + iT + 4 * r * LD];
        };
        auto stC = [&](int zo, const d4s_t(&acc)[NS]) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) Aw[s * SAMP + zo + iT + 4 * r * LD] = acc[s][r];
        };
        auto stCt = [&](int zo, const d4s_t(&acc)[NS]) {  // the accumulator's TRANSPOSE into block zo
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) Aw[s * SAMP + zo + iN + 4 * r] = acc[s][r];
        };
        auto zero = [&](d4s_t(&acc)[NS]) {
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = d4s_t{0, 0, 0, 0};
        };
        double mant = 1.0;
        int ex = 0;
        double *Ad = Aw + (l4 & (NS - 1)) * SAMP;  // the sample this lane's 16-lane row factors
        // ---- potrf
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            s4_diag_block(Ad + s4_blk(j, j), l15, mant, ex);
            S4_STAMP(1);
            if (j + 1 < NB) {
                __builtin_amdgcn_sched_barrier(0);
                // L_ij^T = T_jj A_ij^T for every i > j (T_jj fetched once), kept in registers as pattern N of L_ij
                double lt[NB - 1 > 0 ? NB - 1 : 1][NS][4];
                {
                    double tj[NS][4];
                    ldN(tj, s4_blk(j, j));
#pragma unroll
                    for (int ii = j + 1; ii < NB; ++ii) {
                        double bv[NS][4];
                        ldN(bv, s4_blk(ii, j));
                        d4s_t acc[NS];
                        zero(acc);
                        fmaN(acc, tj, bv, false);
                        stCt(s4_blk(ii, j), acc);  // L_ij itself, for trtri
                        regs(lt[ii - j - 1], acc);
                    }
                }
#pragma unroll
                for (int ii = j + 1; ii < NB; ++ii)
#pragma unroll
                    for (int l = j + 1; l <= ii; ++l) {  // A_il -= L_ij L_lj^T, both operands from registers
                        d4s_t acc[NS];
                        ldC(acc, s4_blk(ii, l));
                        fmaN(acc, lt[ii - j - 1], lt[l - j - 1], true);
                        stC(s4_blk(ii, l), acc);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            S4_STAMP(2);
        }
        // ---- trtri: W = L^-1 in place (diagonal blocks already hold T)
#pragma unroll
        for (int j = NB - 2; j >= 0; --j)
#pragma unroll
            for (int ii = NB - 1; ii > j; --ii) {
                __builtin_amdgcn_sched_barrier(0);
                d4s_t yt[NS];  // Y^T = sum_t L_tj^T W_it^T
                zero(yt);
#pragma unroll
                for (int t = j + 1; t <= ii; ++t) {
                    double av[NS][4], bv[NS][4];
                    ldT(av, s4_blk(t, j));
                    ldN(bv, s4_blk(ii, t));
                    fmaN(yt, av, bv, false);
                }
                double yv[NS][4], tv[NS][4];
                regs(yv, yt);
                ldT(tv, s4_blk(j, j));
                d4s_t w[NS];
                zero(w);
                fmaN(w, yv, tv, true);  // W_ij = -Y T_jj
                stC(s4_blk(ii, j), w);  // (L_ij is no longer needed: the rows above use L_tj with t < i only)
                __builtin_amdgcn_sched_barrier(0);
            }
        S4_STAMP(3);
        // (unconditional -- past the end the indices are clamped -- so that the registers are dead between the image and here;
        //  the loads are older than this group's output stores: waiting for them never waits for a store)
        request(grp + gstride);
        // (the per-sample scalars of this group, requested with the next group's rows: in registers across lauum and z only)
        double xx, wgt_own;
        int m;
        int lane_z = lane;  // (opaque copy: the per-lane global addresses below are formed HERE, not at the top of the group)
        asm volatile("" : "+v"(lane_z));
        const int l4z = lane_z >> 4;
        {
            const int64_t io = i0 + (l4z & (NS - 1));
            const int64_t ic = io < a.n ? io : a.n - 1;
            xx = a.xx[ic];
            m = (int)a.mc[ic];
            wgt_own = a.w ? a.w[ic] : 1.0;
        }
        // ---- lauum: M^-1 = W^T W, lower blocks, in place; row ii ascending, l = ii last.  No product reads what an earlier one
        // of this sweep wrote, so the operands of step n + 1 are requested before the MFMAs of step n.
        {
            constexpr int NSTEP = NB * (NB + 1) * (NB + 2) / 6;  // sum over (ii, l <= ii) of NB - ii steps t
            double av[2][NS][4], bv[2][NS][4];
            d4s_t acc[NS];
            __builtin_amdgcn_sched_barrier(0);
            ldT(av[0], s4_blk(0, 0));
            ldT(bv[0], s4_blk(0, 0));
            sfor<0, NSTEP>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                constexpr S4Step c = s4_lauum_step(NB, n);
                constexpr int ii = c.ii, l = c.l, t = c.t;
                if constexpr (n + 1 < NSTEP) {
                    constexpr S4Step d = s4_lauum_step(NB, n + 1);
                    ldT(av[(n + 1) & 1], s4_blk(d.t, d.ii));
                    ldT(bv[(n + 1) & 1], s4_blk(d.t, d.l));
                }
                if (t == ii) zero(acc);
                fmaN(acc, av[n & 1], bv[n & 1], false);
                if (t == NB - 1) stC(s4_blk(ii, l), acc);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        S4_STAMP(4);
        // ---- z = M^-1 b, lane = (sample, row); the strict upper block part is read through the symmetric entry
        constexpr int NP = (N * NS + 63) / 64;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int q = lane + 64 * u;
            int s = q / N;
            const int r = q - s * N;
            const bool ok = s < NS;
            s = ok ? s : NS - 1;
            const double *As = Aw + s * SAMP;
            const int rb = r >> 4, rl = r & 15;
            double zacc = 0.0;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const bool lower = cb <= rb;
                const int base = lower ? (rb * (rb + 1) / 2 + cb) * BSZ + rl * LD : (cb * (cb + 1) / 2 + rb) * BSZ + rl;
                const int stride = lower ? 1 : LD;
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) zacc = fma(As[base + cc * stride], As[BOFF + 16 * cb + cc], zacc);
            }
            if (ok) Aw[s * SAMP + ZOFF + r] = zacc;
        }
        S4_STAMP(5);
        // ---- traces of the sample of this lane's 16-lane row
        double quad = 0.0, zz = 0.0, tr = 0.0;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int r = l15 + 16 * m;
            const double bb = Ad[BOFF + r], zr = Ad[ZOFF + r], dd = Ad[s4_blk(m, m) + l15 * (LD + 1)];
            if (r < k) {
                quad = fma(bb, zr, quad);
                zz = fma(zr, zr, zz);
                tr += dd;
            }
        }
        quad = row16_sum(quad);
        zz = row16_sum(zz);
        tr = row16_sum(tr);
        {
            const int64_t i = i0 + (l4z & (NS - 1));
            const bool valid = i < a.n && l4z < NS && (lane_z & 15) == 0;
            const double wgt = wgt_own;
            const double logdet = lean_log(mant) + (double)ex * LN_2;  // (the library logarithm's constants were hoisted and spilled)
            const double lk = sample_llk(xx, quad, logdet, s2, lnsig, m, k);
            // Loads and stores share ONE in-order counter on this part: left to the compiler, the first use of the requested rows
            // (top of the next group) waits for vmcnt(0), i.e. for this group's stores.  The rows have had two phases to arrive:
            // wait for them HERE, before the group's first store is issued.
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), expcnt / lgkmcnt untouched
            if (valid) {
                double *sc = a.sc + i * 4;
                if (EM) {
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
        S4_STAMP(6);
        // ---- outputs through the packed index (coalesced): a chunk of table entries, then every sample's reads and stores
        int lane_o = lane;  // (opaque copy: the table entries are READ AGAIN here -- kept from the image phase they sat in scratch)
        asm volatile("" : "+v"(lane_o));
        double wgt_s[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {  // the samples' weights, from the lanes that hold them
            const long long b = __double_as_longlong(wgt_own);
            const int lo = __builtin_amdgcn_readlane((int)b, 16 * s), hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * s);
            wgt_s[s] = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        }
#pragma unroll
        for (int q0 = 0; q0 < NPK; q0 += QC)
            if (64 * q0 < kp) {
                unsigned tv[QC];
#pragma unroll
                for (int j = 0; j < QC && q0 + j < NPK; ++j) {
                    const int e = lane_o + 64 * (q0 + j);
                    tv[j] = tab[e < kp ? e : kp - 1];
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int64_t i = i0 + s;
                    if (i < a.n) {
                        const double *As = Aw + s * SAMP;
                        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(a.G + i * kp, 0, kp * 8, 0x00020000);
                        double ov[QC];
#pragma unroll
                        for (int j = 0; j < QC && q0 + j < NPK; ++j) {
                            const unsigned t = tv[j];
                            const int off = t & 4095, r = (t >> 12) & 63, c = (t >> 18) & 63;
                            ov[j] = EM ? wgt_s[s] * (As[ZOFF + r] * As[ZOFF + c] + s2 * As[off])  // w P = w (z z^T + s2 M^-1)
                                       : s2 * As[off];                                             // Sigma packed
                        }
#pragma unroll
                        for (int j = 0; j < QC && q0 + j < NPK; ++j) {
                            const long long b = __double_as_longlong(ov[j]);
                            __builtin_amdgcn_raw_buffer_store_b64(u2s_t{(unsigned)b, (unsigned)(b >> 32)}, gr, lane_o * 8, (q0 + j) * 512, 0);
                            if (!EM && a.covs) {
                                const int e = lane_o + 64 * (q0 + j);
                                if (e < kp) {
                                    const unsigned t = tv[j];
                                    const int r = (t >> 12) & 63, c = (t >> 18) & 63;
                                    double *cv = a.covs + i * (int64_t)k * k;
                                    cv[r * k + c] = ov[j];
                                    cv[c * k + r] = ov[j];
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t i = i0 + s;
            if (i < a.n) {
                double *bz = a.Bz + i * (k + 1);
                const double zr = Aw[s * SAMP + ZOFF + lane];  // (lane < N)
                if (EM) {
                    if (lane < k) bz[lane] = wgt_s[s] * zr;  // [w z | w]
                    if (lane == 0) bz[k] = wgt_s[s];
                } else if (lane < k) {
                    bz[lane] = zr;
                    if (a.states) a.states[i * k + lane] = zr;
                }
            }
        }
        S4_STAMP(7);
    }
#ifdef S4_TIMING
    if (wave == 0 && lane == 0)
        for (int p = 0; p < 16; ++p) atomicAdd(&s4_dbg[p], tacc_[p]);
#endif
}

template <int NB, bool EM>
hipError_t launch_solve4_t(const SolveArgs &a, int n_cu, hipStream_t s) {
    constexpr int N = 16 * NB, W = s4_waves(NB), NS = s4_ns(NB);
    const size_t lds = sizeof(double) * W * NS * s4_samp(NB) + sizeof(unsigned) * (N * (N + 1) / 2);
    const int64_t groups = (a.n + NS - 1) / NS;
    int grid = (int)std::min<int64_t>((groups + W - 1) / W, (int64_t)n_cu);
    if (grid < 1) grid = 1;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&solve4_kernel<NB, EM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((solve4_kernel<NB, EM>), dim3(grid), dim3(64 * W), lds, s, a);
    return hipGetLastError();
}

}  // namespace

bool solve4_covers(int k) { return k > 16 && k <= 64; }

hipError_t launch_solve4(const SolveArgs &a, int n_cu, hipStream_t s) {
    if (a.em) {
        if (a.k <= 32) return launch_solve4_t<2, true>(a, n_cu, s);
        if (a.k <= 48) return launch_solve4_t<3, true>(a, n_cu, s);
        return launch_solve4_t<4, true>(a, n_cu, s);
    }
    if (a.k <= 32) return launch_solve4_t<2, false>(a, n_cu, s);
    if (a.k <= 48) return launch_solve4_t<3, false>(a, n_cu, s);
    return launch_solve4_t<4, false>(a, n_cu, s);
}

}  // namespace ppca
