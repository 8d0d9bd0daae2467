// ppca_em16.hip -- the EM pass for state sizes 11 .. 16 at d <= 256 (the reference's own largest workload, lib.rs:82-99:
// d = 200, k = 16) as TWO fused kernels with the rows of [wP | wz | w] handed over through HBM:
//
//   estep16_kernel<K>   one sweep over X: staging (centring, masks), [G | b] (int8-sliced Gram + fp64-MFMA b = X~ C),
//                       the per-sample k x k solve, the x~-side statistics cross / sumx, the scalars and the llk --
//                       the front role of em8_kernel (ppca_em8.hip) as its own four-wave workgroup with the whole
//                       512-entry register file per wave; from k = 14 the packed Cholesky factor (210+ registers) is split
//                       over lane pairs and parked in LDS for the substitutions (SplitChol, solve_lds, minv_pair_lds);
//                       writes each sample's row [wP (k') | wz (k) | w] (fp64) and the tile's per-dimension sample masks
//   sstat16_kernel<K>   S / U / totals (256 x (k' + k + 1)) += Mask^T [wP | wz | w] on the INT8 MFMA with exact 64-bit
//                       integer accumulation: the back role of em8_kernel as its own eight-wave workgroup (each wave owns
//                       32 dimensions x all columns: 160 accumulator registers at k = 16), rows staged from HBM
//
// Why two kernels: at k = 16 the mask-side statistics are 256 x 153 columns -- 320 accumulator registers per wave of a
// four-wave back role, and the digit planes of a 64-sample group (71 KB) do not fit beside the x~ tile, C and [G | b] in
// the 160 KB of LDS.  Through HBM the hand-over costs 1 280 B written + read per sample next to the 2 048 B of the row of
// X itself; each kernel then has the whole CU.  Before this kernel the shapes ran on the split pipeline of
// ppca_generic.hip (ten launches per chunk, X read four times): 14.5 ms per iteration at d = 200, k = 16, N = 2 M;
// here 6.7 ms (profiles/r03/cliff.md).
//
// Fixed-point form, violation handling and flush cadence of the contraction: exactly as in ppca_em8.hip (one exponent
// per column and workgroup, set by the first tile + 6 binary orders of head room; a tile that does not fit flushes the
// integers and raises the exponents).
//
// Replaces in the reference: infer (ppca/src/ppca_model.rs:221-227), the cross moment (:281-293), the d second-moment
// scans (:294-306), the noise 4-tuple (:328-358) and llk (:142-149), for 11 <= k <= 16.
#include <atomic>
#include <cstdlib>

#include "ppca_device.hpp"

#ifndef E16_X_AUX
#define E16_X_AUX 0  // cache policy of the row loads of X (2 = nt: read once, served past the L1 that C and the digit table live in)
#endif
namespace ppca {

constexpr int E16_QW = 7;        // signed bytes per entry
constexpr int E16_F = 50;        // |I| < 2^F
constexpr int E16_HEAD = 6;      // binary orders kept free above the column maximum of the tile that set the scale
constexpr int E16_POISON = 100000;
constexpr int E16_EMIN = -900, E16_EMAX = 1000;
constexpr int E16_FLUSH_GROUPS = 100;
constexpr int E16_MIN_K = 11, E16_MAX_K = 16;
#ifndef E16_SPLIT_MIN_K
#define E16_SPLIT_MIN_K 14
#endif

template <int K>
struct Cfg16 {
    static constexpr int KP = K * (K + 1) / 2;
    static constexpr int NTP = (KP + 15) / 16;
    static constexpr int B = 32, DP = 256, XS = DP + 2;
    // SPLIT: the per-sample solve with the factor split over lane pairs and parked in LDS (k >= 14: 105+ packed doubles
    // and the solver's working set no longer fit a lane's 256 directly addressable registers) -- then C is read from
    // L2 per tile and its place in LDS holds the factors; else the factor lives in registers and C in LDS
    static constexpr bool SPLIT = K >= E16_SPLIT_MIN_K;
    static constexpr int CS = K + 1;             // C tile row stride; column K is all zeros
    static constexpr int LS = KP | 1;            // row stride of the factor buffer (odd: conflict-free per-lane rows)
    static constexpr int NC = KP + K + 1;        // statistic columns [wP | wz | w]
    static constexpr int NCT = (NC + 15) / 16;
    static constexpr int NCOL = 16 * NCT;        // row stride of the hand-over buffer (doubles)
    // ---- estep16_kernel
    // [G | b] and the [wP | wz | w] rows share ONE buffer (row stride GS), as in em8_kernel:
    //   as [G | b]:  G (16 NTP, K' used) | b partial of dims 0-127 (16) | pad
    //   as W row:    wP (K') | wz (K) | w | ..      (compact: the hand-over buffer's row)
    static constexpr int GS = 16 * NTP + 18;
    static_assert(GS % 2 == 0 && GS >= NCOL, "W rows: 16-byte pieces, compact [wP | wz | w | pad] inside the row");
    static constexpr int BS = K + 1;             // b partial of dims 128-255
    static constexpr int OFF_X = 0;
    static constexpr int OFF_LB = OFF_X + B * XS;         // packed Cholesky factors of the tile's samples
    static constexpr int OFF_C = OFF_LB;                  // ... or C (not SPLIT)
    static constexpr int OFF_G = OFF_LB + (SPLIT ? B * LS : DP * CS);
    static constexpr int OFF_B1 = OFF_G + B * GS;
    static constexpr int OFF_M = OFF_B1 + B * BS;         // mask words, two parities x B x 4 u64
    static constexpr int OFF_MB = OFF_M + 2 * B * 4;      // sample masks per dimension of the staged tile: DP u32
    static constexpr int OFF_S = OFF_MB + DP / 2;         // cross-wave scratch
    static constexpr int OFF_L = OFF_S + 2 * B;           // running scalars
    static constexpr int OFF_MU = OFF_L + 14 * B;         // the mean (DP doubles, zero past d)
    static constexpr int OFF_K = OFF_MU + DP;             // model scalars
    static constexpr int LDS_DOUBLES = OFF_K + 4;
    static_assert(LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget (estep16)");
    // ---- sstat16_kernel
    static constexpr int PLANE_BYTES = E16_QW * 2 * NCOL * 16;  // digit planes of one tile: [plane][16-sample chunk][column][16 B]
    static constexpr int S_OFF_R = 0;                            // two staged tiles of rows: 2 x B x NCOL doubles
    static constexpr int S_OFF_P0 = S_OFF_R + 2 * B * NCOL;
    static constexpr int S_OFF_P1 = S_OFF_P0 + PLANE_BYTES / 8;
    static constexpr int S_OFF_MB = S_OFF_P1 + PLANE_BYTES / 8;  // DP x 4 u32 (slot = tile % 3)
    static constexpr int S_OFF_E = S_OFF_MB + DP * 2;            // column exponents
    static constexpr int S_OFF_BAR = S_OFF_E + NCOL / 2 + 4;
    static constexpr int S_LDS_DOUBLES = S_OFF_BAR + 4;
    static_assert(S_LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget (sstat16)");
    static_assert(NCOL / 4 <= 64, "one wave digitises NCOL / 4 (column, chunk) items");
};

struct S16Args {
    const double *Wrows;
    const unsigned *Mb;
    int64_t n;
    int d;
    double *part;          // [grid][stats_len]: this kernel writes S, U, totals
};

#ifndef E16_FACTOR_SWAP
#define E16_FACTOR_SWAP 0
#endif
#ifdef PPCA_PHASE_TIMING
#define E16_STAMP(i) { long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; }
#else
#define E16_STAMP(i)
#endif

// ------------------------------------------------------------------ the per-sample solve of estep16_kernel
// Broadcast of one half of the wave to both: v_permlane32_swap with both operands the same register leaves
// (lower | lower) in the first result and (upper | upper) in the second.
template <int H>
__device__ __forceinline__ double bcast_half(double x) {
    const long long b = __double_as_longlong(x);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((long long)hi[H] << 32) | lo[H]);
}

// Cholesky factor of M = G + s2 I, split over the lane pair (lane, lane ^ 32): half h = lane >> 5 owns the columns
// b = 2 j + h.  Slot (a, j), j <= a / 2, holds entry (a, 2 j + h) (half 1's slot (a, a / 2) of an even row a is unused
// and never read).  Right-looking: per pivot column c its owner's entries are broadcast to both halves (two
// v_permlane32_swap per double), then every half updates its own columns -- all multiply-adds independent.  The same
// arithmetic per entry as Posterior<K>::factor_loaded (ppca_small.hpp): diagonal slots keep 1 / L_aa.
template <int K>
struct SplitChol {
    static constexpr int off(int a) {
        int s = 0;
        for (int r = 0; r < a; ++r) s += r / 2 + 1;
        return s;
    }
    static constexpr int idx(int a, int j) { return off(a) + j; }
    static constexpr int NS = off(K);
    double L[NS];

    __device__ __forceinline__ void load(const double *g, int hi, double s2) {
        const double *gh = g + hi;
        static_for<K>([&](auto a_tag) {
            constexpr int a = decltype(a_tag)::value;
            static_for<a / 2 + 1>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                double v = gh[tri(a, 2 * j)];  // entry (a, 2 j + hi): neighbours in the packed row
                if constexpr (2 * j == a) v = hi ? 0.0 : v + s2;      // diagonal of an even row (half 0); half 1: unused slot
                else if constexpr (2 * j + 1 == a) v = hi ? v + s2 : v;  // diagonal of an odd row (half 1)
                L[idx(a, j)] = v;
            });
        });
    }
    __device__ __forceinline__ void factor(int hi, double &pm, int &pe) {
        double grp[2] = {1.0, 1.0};
        static_for<K>([&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value, hc = c & 1, jc = c >> 1;
            const bool own = hi == hc;
            const double piv = bcast_half<hc>(L[idx(c, jc)]);
            grp[c >= (K + 1) / 2] *= piv;
            const double inv = fast_rsqrt(piv);  // 1 / L_cc
            L[idx(c, jc)] = own ? inv : L[idx(c, jc)];
            if constexpr (c + 1 < K) {
                double pc[K];
#pragma unroll
                for (int a = c + 1; a < K; ++a) {
                    pc[a] = bcast_half<hc>(L[idx(a, jc)]) * inv;
                    L[idx(a, jc)] = own ? pc[a] : L[idx(a, jc)];
                }
                // second factors of this half's columns: column 2 j + hi (the pivot column itself, in half 0 of an even
                // step, gets a zero: its slots hold the scaled column already)
                constexpr int jlo = hc ? jc + 1 : jc, jhi = (K - 1) / 2;
                double ps[jhi + 1];
#pragma unroll
                for (int j = jlo; j <= jhi; ++j) {
                    const double v0 = (2 * j > c) ? pc[2 * j] : 0.0;
                    const double v1 = (2 * j + 1 < K) ? pc[2 * j + 1] : 0.0;
                    ps[j] = hi ? v1 : v0;
                }
#pragma unroll
                for (int a = c + 1; a < K; ++a)
#pragma unroll
                    for (int j = jlo; j <= a / 2; ++j) L[idx(a, j)] -= pc[a] * ps[j];
            }
        });
        int e0, e1;
        pm = frexp(grp[0], &e0) * frexp(grp[1], &e1);
        pe = e0 + e1;
    }
    // The same factorisation with the pivot column exchanged through LDS instead of v_permlane32_swap: the owner half
    // drops the raw column into the wave's exchange slots xch[0 .. K) (a dead stretch of the sample's Gram row), both
    // halves read it back (one address per lane pair), scale it themselves, and the scaled column -- final -- goes
    // straight to the factor buffer lrow (wave c mod 4 writes column c: the four waves hold the same values).  A swap
    // cost two copies and two wait states per word on top of itself (the instruction overwrites both operands), and the
    // owner's registers had to be refreshed by selects: 2 200 instructions per factorisation against ~1 100 here.
    __device__ __forceinline__ void factor_lds(int hi, double *xch, double *lrow, int wave, double &pm, int &pe) {
        double grp[2] = {1.0, 1.0};
        double raw[2][K];  // the pivot column as read back, current / next step
        if (hi == 0) {
#pragma unroll
            for (int a = 0; a < K; ++a) xch[a] = L[idx(a, 0)];
        }
        // (lanes exchange data here: the loads must stay behind the other half's stores -- to a single thread's memory
        //  model they are independent; the LDS itself serves a wave's operations in order)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int a = 0; a < K; ++a) raw[0][a] = xch[a];
        static_for<K>([&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value, hc = c & 1, jc = c >> 1;
            double pc[K];
            const double piv = raw[c & 1][c];
            grp[c >= (K + 1) / 2] *= piv;
            const double inv = fast_rsqrt(piv);  // 1 / L_cc
#pragma unroll
            for (int a = c + 1; a < K; ++a) pc[a] = raw[c & 1][a] * inv;
            if constexpr (c + 1 < K) {
                constexpr int jlo = hc ? jc + 1 : jc, jhi = (K - 1) / 2;
                double ps[jhi + 1];
#pragma unroll
                for (int j = jlo; j <= jhi; ++j) {
                    const double v0 = (2 * j > c) ? pc[2 * j] : 0.0;
                    const double v1 = (2 * j + 1 < K) ? pc[2 * j + 1] : 0.0;
                    ps[j] = hi ? v1 : v0;
                }
                // the next pivot column first (column c + 1 = slot j1 of half h1): it leaves for the exchange slots and
                // is requested back at once, and the round trip runs under the rest of this step's updates
                constexpr int c1 = c + 1, h1 = c1 & 1, j1 = c1 >> 1;
#pragma unroll
                for (int a = c1; a < K; ++a) L[idx(a, j1)] -= pc[a] * ps[j1];
                if (hi == h1) {
#pragma unroll
                    for (int a = c1; a < K; ++a) xch[a] = L[idx(a, j1)];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int a = c1; a < K; ++a) raw[c1 & 1][a] = xch[a];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = c + 1; a < K; ++a)
#pragma unroll
                    for (int j = jlo; j <= a / 2; ++j)
                        if (j != j1) L[idx(a, j)] -= pc[a] * ps[j];
            }
            if ((c & 3) == wave && hi == hc) {  // the scaled column is final: into the factor buffer
                lrow[tri(c, c)] = inv;
#pragma unroll
                for (int a = c + 1; a < K; ++a) lrow[tri(a, c)] = pc[a];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        int e0, e1;
        pm = frexp(grp[0], &e0) * frexp(grp[1], &e1);
        pe = e0 + e1;
    }
    // the factor into the packed row (the four waves hold the same values: each writes the rows a = wave mod 4)
    __device__ __forceinline__ void store(double *g, int hi, int wave) const {
        double *gh = g + hi;
        static_for<K>([&](auto a_tag) {
            constexpr int a = decltype(a_tag)::value;
            if ((a & 3) == wave) {
                static_for<a / 2 + 1>([&](auto j_tag) {
                    constexpr int j = decltype(j_tag)::value;
                    if constexpr (2 * j == a) {
                        if (!hi) gh[tri(a, 2 * j)] = L[idx(a, j)];
                    } else {
                        gh[tri(a, 2 * j)] = L[idx(a, j)];
                    }
                });
            }
        });
    }
};

// The column pairs of M^-1 of worker w of 4 (every worker also solves for z, so the pairs alone are balanced): pair w
// and its mirror 7 - w -- the costs, (K - 2 p)^2, fall with p, so heavy goes with light; -1: none.
constexpr int first_pair(int w) { return w; }
constexpr int second_pair(int K, int w) { return 7 - w < (K + 1) / 2 ? 7 - w : -1; }

// z = M^-1 b (b in z), quad = b^T M^-1 b, zz = |z|^2 with the packed factor read from LDS (Posterior<K>::solve_loaded).
// A substitution is a chain of K dependent steps, each needing one column (forward) or row (backward) of the factor:
// the entries of the NEXT step are requested before the current step's arithmetic, so a step costs its multiply-adds,
// not an LDS round trip (measured before: ~150 cycles per step, 32 steps per solve).
template <int K>
__device__ __forceinline__ void solve_lds(const double *Lr, double (&z)[K], double &quad, double &zz) {
    double buf[2][K];
#pragma unroll
    for (int a = 0; a < K; ++a) buf[0][a] = Lr[tri(a, 0)];
    quad = 0.0;
    static_for<K>([&](auto t_tag) {
        constexpr int t = decltype(t_tag)::value;
        if constexpr (t + 1 < K) {
#pragma unroll
            for (int a = t + 1; a < K; ++a) buf[(t + 1) & 1][a] = Lr[tri(a, t + 1)];
        } else {
#pragma unroll
            for (int a = 0; a < K; ++a) buf[(t + 1) & 1][a] = Lr[tri(K - 1, a)];  // the first row of the backward pass
        }
        __builtin_amdgcn_sched_barrier(0);
        z[t] *= buf[t & 1][t];
        quad += z[t] * z[t];
#pragma unroll
        for (int a = t + 1; a < K; ++a) z[a] -= buf[t & 1][a] * z[t];
        __builtin_amdgcn_sched_barrier(0);
    });
    zz = 0.0;
    static_for<K>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value, t = K - 1 - s, pb = (K + s) & 1;
        if constexpr (t >= 1) {
#pragma unroll
            for (int a = 0; a < t; ++a) buf[pb ^ 1][a] = Lr[tri(t - 1, a)];
        }
        __builtin_amdgcn_sched_barrier(0);
        z[t] *= buf[pb][t];
        zz += z[t] * z[t];
#pragma unroll
        for (int a = 0; a < t; ++a) z[a] -= buf[pb][a] * z[t];
        __builtin_amdgcn_sched_barrier(0);
    });
}
// Columns C0 (half 0) and C0 + 1 (half 1) of M^-1, rows t >= C0, factor read from LDS with the same one-step-ahead
// requests (Posterior<K>::minv_column_pair); st(t, v): entry (t, C0 + hi).  Returns (M^-1) at (C0 + hi, C0 + hi).
template <int K, int C0, class Store>
__device__ __forceinline__ double minv_pair_lds(const double *Lr, int hi, Store st) {
    double u[K];
#pragma unroll
    for (int a = C0; a < K; ++a) u[a] = (a == C0 + hi) ? 1.0 : 0.0;
    double buf[2][K];
#pragma unroll
    for (int a = C0; a < K; ++a) buf[C0 & 1][a] = Lr[tri(a, C0)];
    static_for<K - C0>([&](auto s_tag) {
        constexpr int t = C0 + decltype(s_tag)::value;
        if constexpr (t + 1 < K) {
#pragma unroll
            for (int a = t + 1; a < K; ++a) buf[(t + 1) & 1][a] = Lr[tri(a, t + 1)];
        } else {
#pragma unroll
            for (int a = C0; a < K; ++a) buf[(t + 1) & 1][a] = Lr[tri(K - 1, a)];
        }
        __builtin_amdgcn_sched_barrier(0);
        u[t] *= buf[t & 1][t];
#pragma unroll
        for (int a = t + 1; a < K; ++a) u[a] -= buf[t & 1][a] * u[t];
        __builtin_amdgcn_sched_barrier(0);
    });
    double diag = 0.0;
    static_for<K - C0>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value, t = K - 1 - s, pb = (K + s) & 1;
        if constexpr (t - 1 >= C0) {
#pragma unroll
            for (int a = C0; a < t; ++a) buf[pb ^ 1][a] = Lr[tri(t - 1, a)];
        }
        __builtin_amdgcn_sched_barrier(0);
        u[t] *= buf[pb][t];
        st(t, u[t]);
        if constexpr (t == C0 + 1) diag = hi ? u[t] : diag;
        if constexpr (t == C0) diag = hi ? diag : u[t];
#pragma unroll
        for (int a = C0; a < t; ++a) u[a] -= buf[pb][a] * u[t];
        __builtin_amdgcn_sched_barrier(0);
    });
    return diag;
}

// ===================================================================== E-step
template <int K, bool WEIGHTED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void estep16_kernel(Em16Launch p) {
    using cfg = Cfg16<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, B = cfg::B, XS = cfg::XS, GS = cfg::GS, BS = cfg::BS, LS = cfg::LS,
                  WS = cfg::GS, NC = cfg::NC, NCOL = cfg::NCOL;
    constexpr bool SPLIT = cfg::SPLIT;
    constexpr int CS = cfg::CS;
    constexpr int NF = 4;
    constexpr int RPW = B / NF;        // rows staged per wave
    constexpr int DPS = cfg::DP / 2;   // dims per K-split of b = X~ C
    constexpr int STEPS = DPS / 4;
    constexpr int RT = 16 / NF;        // 16-dim row tiles per wave in P4a
    constexpr int DW = cfg::DP / NF;
    constexpr int MAXT = (NTP + NF - 1) / NF;  // packed-column tiles of the Gram per wave
    static_assert(QS == 8, "digit grouping assumes 8 slices");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Lb = sm + cfg::OFF_LB;
    double *Cs = sm + cfg::OFF_C;
    double *Gs = sm + cfg::OFF_G;
    double *B1 = sm + cfg::OFF_B1;
    double *Ws = sm + cfg::OFF_G;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    unsigned *Mb = reinterpret_cast<unsigned *>(sm + cfg::OFF_MB);
    double *xxs = sm + cfg::OFF_S;
    double *scl = sm + cfg::OFF_L;

    const int tid = threadIdx.x, lane_entry = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = p.d;
    const int64_t n = p.n;
    const double *mC = p.model + MODEL_HDR;
    const double *mMean = mC + (int64_t)d * K;
    const double s2_k = p.model[1], lnsig_k = p.model[2];
    int unsafe = 0;
#pragma unroll
    for (int t = 0; t < NTP; ++t) unsafe |= p.qflag[t];
    unsafe = __builtin_amdgcn_readfirstlane(unsafe);

    if constexpr (!SPLIT) {
        for (int idx = tid; idx < cfg::DP * CS; idx += 256) {
            int j = idx / CS, a = idx - j * CS;
            Cs[idx] = (j < d && a < K) ? mC[(int64_t)j * K + a] : 0.0;
        }
    }
    constexpr int SQW = 2 * B;  // sq slots per wave (lane pairs)
    constexpr int L_DEV = NF * SQW, L_LLK = L_DEV + B, L_W = L_DEV + 2 * B, L_NE = L_DEV + 3 * B, L_PM = L_DEV + 4 * B,
                  L_PX = L_DEV + 5 * B;
    static_assert(L_DEV + 6 * B <= 14 * B, "scalar slots");
    for (int idx = tid; idx < L_DEV + 6 * B; idx += 256) scl[idx] = (idx >= L_PM && idx < L_PX) ? 1.0 : 0.0;
    for (int idx = tid; idx < cfg::DP; idx += 256) {
        Mb[idx] = 0u;
        sm[cfg::OFF_MU + idx] = idx < d ? mMean[idx] : 0.0;
    }
    if (tid == 0) {
        sm[cfg::OFF_K] = s2_k;
        sm[cfg::OFF_K + 1] = 1.0 / s2_k;
        sm[cfg::OFF_K + 2] = lnsig_k;
    }

    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
#ifdef PPCA_PHASE_TIMING
    long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
#endif
    __syncthreads();

    // cross accumulators on v_mfma_f64_4x4x4 (four 4 x 4 x 4 blocks per instruction: block b = lane bits 2-3; A[i][k] in lane
    // 16 k + 4 b + i, B[k][j] in lane 16 k + 4 b + j, D[i][j] in lane 16 i + 4 b + j): one wave issues it every 18 cycles for 512 flop
    // where v_mfma_f64_16x16x4 takes 142 for 2 048 (profiles/r04/mfma_peak.txt); the x~ operand registers are the same
    constexpr int NCG = (K + 3) / 4;
    double accX[RT][NCG];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NCG; ++c) accX[r][c] = 0.0;
    const double inv_s2_k = 1.0 / s2_k;
    double xr[RPW][4];
    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    const int64_t own = (tile_end - tile_begin) * B;
    const int nmine = tile_end > tile_begin ? (int)(own < nleft ? own : nleft) : 0;
    const int rowbytes = (int)p.ldx * (int)sizeof(double);  // (lanes past d of a row read its neighbour: masked by lim)
    auto tile_rsrc = [&](int64_t tile) {
        const int rel0 = (int)(tile - tile_begin) * B;
        int cnt = nrel - rel0;
        cnt = __builtin_amdgcn_readfirstlane(cnt < 0 ? 0 : (cnt > B ? B : cnt));
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Xwg + (int64_t)rel0 * p.ldx), 0, cnt * rowbytes, 0x00020000);
    };
    auto load_row = [&](const __amdgpu_buffer_rsrc_t &trs, int r) {
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(trs, lane_entry * 16, (wave * RPW + r) * rowbytes + 1024 * h, E16_X_AUX);
            xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
            xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
        }
    };
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, NTP * QS * 4 * 1024, 0x00020000);
    auto load_pair = [&](i4_t(&dst)[2][4], int ti, int sl0) {
        int qbase = (ti * QS + sl0) * 4096;
        asm volatile("" : "+s"(qbase));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16 + kc * 1024, qbase + u * 4096, 0);
                dst[u][kc] = i4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
    };
    // ---- P1: as em8_kernel (the finite-test ballots ARE the mask words; each lane shifts its own bit of every ballot
    // into st_mb: byte = this dimension over the wave's eight samples, row r at bit 7 - r); sumx rides along on the
    // vector unit (four multiply-adds per row: the sixteen columns of the cross product's MFMA tile are all wz at k = 16)
    double mu[4], lim[4];
    int st_wlo = 0, st_whi = 0;
    int st_mb[4] = {0, 0, 0, 0};
    double xx_run = 0.0;
    double sx_run[4] = {0.0, 0.0, 0.0, 0.0};
    auto stage_row = [&](int64_t t, int lane, auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        const int ri = wave * RPW + r;
        const bool mine = (int)(t - tile_begin) * B + ri < nmine;
        const double wr = mine ? (WEIGHTED ? p.w[t * B + ri] : 1.0) : 0.0;  // (scalar load)
        double pc_xx = 0.0;
        static_for<2>([&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value;
            double xt[2];
            unsigned long long bal[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double v = xr[r][2 * h + e];
                const bool ob = __builtin_fabs(v) < lim[2 * h + e];
                bal[e] = __builtin_amdgcn_ballot_w64(ob);
                xt[e] = ob ? v - mu[2 * h + e] : 0.0;  // select, never multiply (utils.rs:118-127)
            }
            file_mask<4 * r + 2 * h>(st_wlo, st_whi, st_mb[2 * h], bal[0]);
            file_mask<4 * r + 2 * h + 1>(st_wlo, st_whi, st_mb[2 * h + 1], bal[1]);
            typedef double d2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 * h + 2 * lane) = d2_t{xt[0], xt[1]};
            pc_xx += xt[0] * xt[0];
            pc_xx += xt[1] * xt[1];
            sx_run[2 * h] += wr * xt[0];
            sx_run[2 * h + 1] += wr * xt[1];
        });
        xx_run += wr * pc_xx;
    };
    auto stage_tile = [&](int64_t t, int lane) {
        {
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t m0 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 2 * lane);
            const d2_t m1 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 128 + 2 * lane);
            mu[0] = m0[0]; mu[1] = m0[1]; mu[2] = m1[0]; mu[3] = m1[1];
#pragma unroll
            for (int q = 0; q < 4; ++q) lim[q] = (128 * (q >> 1) + 2 * lane + (q & 1) < d) ? __builtin_inf() : -1.0;
        }
        const int rel = (int)(t - tile_begin);
        st_wlo = st_whi = 0;
        st_mb[0] = st_mb[1] = st_mb[2] = st_mb[3] = 0;
        static_for<RPW>([&](auto r_tag) { stage_row(t, lane, r_tag); });
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[(rel & 1) * 4 * B + wave * 4 * RPW + lane] = myw;
        unsigned char *mbb = reinterpret_cast<unsigned char *>(Mb);
#pragma unroll
        for (int q = 0; q < 4; ++q) mbb[(128 * (q >> 1) + 2 * lane + (q & 1)) * 4 + wave] = (unsigned char)st_mb[q];
    };

    // (tried: a quarter-tile-period start offset per workgroup against bursts of the chip's memory requests -- every
    //  workgroup does the same work per tile -- : no effect, the stall it was aimed at was a scratch reload's vmcnt(0))
    // B operand of b = X~ C: this wave's K-half of C (dims DPS kq + l4 + 4 u, column l15; zeros past d and past K).  The
    // fragments do not depend on the tile, but as kernel-long register residents they were what the factorisation spilled
    // (and their place in LDS holds the factors): they are requested from L2 behind the previous tile's P4a, arrive
    // during the staging and die with the b product at the head of P2.
    double cf[STEPS];
    // (one buffer descriptor over C: the lane's offset is one register, the k-step a scalar; dims past d read as zeros,
    //  lanes past column K are sent out of range)
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(mC), 0, d * K * (int)sizeof(double), 0x00020000);
    auto load_cf = [&](int lane) {
        const int l15 = lane & 15, l4 = lane >> 4, kq = wave >> 1;
        const int voff = l15 < K ? ((DPS * kq + l4) * K + l15) * 8 : 0x7FFFFFF0;
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            typedef unsigned u2_t __attribute__((ext_vector_type(2)));
            const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(crsrc, voff, 4 * u * K * 8, 0);
            cf[u] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
        }
    };
    if (tile_begin < tile_end) {
        const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile_begin);
#pragma unroll
        for (int r = 0; r < RPW; ++r) load_row(trs, r);
        if constexpr (SPLIT) load_cf(lane_entry);
        stage_tile(tile_begin, lane_entry);
    }
    __syncthreads();

    for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
        const int rel = (int)(tile - tile_begin);
        const unsigned long long *Msc = Ms + (rel & 1) * 4 * B;
        // ------------------------------------------------------------ P2: [G | b] of the tile
        {
            const int rt = wave & 1, kq = wave >> 1;
            const int si = 16 * rt + l15;
            d4_t accb = d4_t{0, 0, 0, 0};
            const double *xrow = Xs + si * XS + DPS * kq + l4;
            i4_t qb[2][2][4];
            if (!unsafe) load_pair(qb[0], wave, 6);  // (every wave owns at least one tile: NTP >= 4)
            if constexpr (SPLIT) {
                // b = X~ C: the A operands of the next four k-steps are requested before the current four MFMAs issue
                constexpr int CH = 4;
                double axb[2][CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) axb[0][u] = xrow[4 * u];
#pragma unroll
                for (int c = 0; c < STEPS / CH; ++c) {
                    if (c + 1 < STEPS / CH) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) axb[(c + 1) & 1][u] = xrow[4 * ((c + 1) * CH + u)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CH; ++u) accb = mfma(axb[c & 1][u], cf[c * CH + u], accb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                constexpr int CH = 4;
                const double *cpc = Cs + (DPS * kq + l4) * CS + (l15 < K ? l15 : K);
                double axb[2][CH], cbb[2][CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    axb[0][u] = xrow[4 * u];
                    cbb[0][u] = cpc[4 * u * CS];
                }
#pragma unroll
                for (int c = 0; c < STEPS / CH; ++c) {
                    if (c + 1 < STEPS / CH) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            axb[(c + 1) & 1][u] = xrow[4 * ((c + 1) * CH + u)];
                            cbb[(c + 1) & 1][u] = cpc[4 * ((c + 1) * CH + u) * CS];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CH; ++u) accb = mfma(axb[c & 1][u], cbb[c & 1][u], accb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            E16_STAMP(0)
            if (!unsafe) {
                // int8-sliced Gram: A = mask bytes (lane: sample 16 rt2 + l15, k-chunk kc, dims 16 l4 .. +15 of it), B = the
                // digit table of packed-column tile ti; tiles wave, wave + 4, wave + 8
                i4_t af[2][4];
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) {
                        const unsigned bits = (unsigned)(Msc[(16 * rt2 + l15) * 4 + kc] >> (16 * l4)) & 0xFFFFu;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            af[rt2][kc][u] = (int)((((bits >> (4 * u)) & 0xFu) * 0x00204081u) & 0x01010101u);
                    }
                double v[2][4];
                static_for<MAXT * 4>([&](auto s_tag) {
                    constexpr int s = decltype(s_tag)::value, tt = s / 4, g = 3 - s % 4;
                    const int ti = wave + NF * tt;
                    if (ti < NTP) {  // (wave-uniform)
                        if constexpr (s + 1 < MAXT * 4) {
                            constexpr int tt1 = (s + 1) / 4, g1 = 3 - (s + 1) % 4;
                            if (wave + NF * tt1 < NTP) load_pair(qb[(s + 1) & 1], wave + NF * tt1, 2 * g1);
                        }
                        // one digit pair: contract, then fold the exact integer digit sums into the running fp64 value
#pragma unroll
                        for (int rt2 = 0; rt2 < 2; ++rt2) {
                            i4_t ia[2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                ia[u] = i4_t{0, 0, 0, 0};
#pragma unroll
                                for (int kc = 0; kc < 4; ++kc)
                                    ia[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rt2][kc], qb[s & 1][u][kc], ia[u], 0, 0, 0);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int part = ia[1][r] * QBASE + ia[0][r];
                                v[rt2][r] = (g == 3) ? (double)part : v[rt2][r] * (double)(QBASE * QBASE) + (double)part;
                            }
                        }
                        if constexpr (g == 0) {
                            const double qs = p.qscale[16 * ti + l15];
#pragma unroll
                            for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                                for (int r = 0; r < 4; ++r)  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                                    Gs[(16 * rt2 + 4 * l4 + r) * GS + 16 * ti + l15] = v[rt2][r] * qs;
                        }
                    }
                });
            } else {
                // the guard tripped: packed Gram rows of the fp64 engine (a guarded launch before this kernel)
                const int64_t row0 = tile * B;
                for (int idx = 64 * wave + lane; idx < B * KP; idx += 256) {
                    const int r = idx / KP, e = idx - r * KP;
                    Gs[r * GS + e] = (row0 + r < n) ? p.Gext[(row0 + r) * KP + e] : 0.0;
                }
            }
            E16_STAMP(1)
            // the two K-split partials of b are summed by the solver in a fixed order (p0 + p1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kq == 0) Gs[(16 * rt + l4 + 4 * r) * GS + 16 * NTP + l15] = accb[r];
                else if (l15 < K + 1) B1[(16 * rt + l4 + 4 * r) * BS + l15] = accb[r];
            }
        }
        __syncthreads();
        E16_STAMP(2)
        // ------------------------------------------------------------ P3: per-sample k x k solve
        // The packed Cholesky factor (136 doubles at k = 16) does not fit the 256 directly addressable registers of a
        // lane: a lane-per-sample factorisation ran on register-file copies (9 x the arithmetic).  Here the lane pair
        // (i, i + 32) holds the factor column-cyclically (SplitChol: 72 doubles each) and exchanges the pivot column
        // per step; the finished factor goes to its own LDS buffer, and the substitutions -- z in every wave, the
        // columns of M^-1 shared by the waves, two per instruction stream as in em8_kernel -- read it from there
        // (same address in both halves: one broadcast read), writing the W rows over the Gram as they go.
        {
            const double s2 = sm[cfg::OFF_K], inv_s2 = sm[cfg::OFF_K + 1], lnsig = sm[cfg::OFF_K + 2];
            const int i = lane & (B - 1);
            const int hi = lane >> 5;
            const int64_t row = tile * B + i;
            const double *g0 = Gs + i * GS;
            double *b1 = B1 + i * BS;
            double *lrow = Lb + i * LS;
            const double wgt = (row < n) ? (WEIGHTED ? p.w[row] : 1.0) : 0.0;
            const int m = __popcll(Msc[i * 4]) + __popcll(Msc[i * 4 + 1]) + __popcll(Msc[i * 4 + 2]) + __popcll(Msc[i * 4 + 3]);
            double *wrow = Ws + i * WS;
            double sc_sq = 0.0, sc_dev = 0.0, sc_llk = 0.0, sc_w = 0.0, sc_ne = 0.0;
            double trpart = 0.0;
            double pm;
            int pe;
            double z[K], quad = 0.0, zz = 0.0;
            if constexpr (SPLIT) {
            {
                SplitChol<K> sc;
                sc.load(g0, hi, s2);
                if (wave == 0 && hi == 0) {  // b = p0 + p1 into the second partial's slots: the first's are W-row ground
#pragma unroll
                    for (int a = 0; a < K; ++a) b1[a] = g0[16 * NTP + a] + b1[a];
                }
                __syncthreads();  // every wave holds its operands: the W rows go where [G | b] is
                E16_STAMP(3)
#if E16_FACTOR_SWAP  // (the exchange by v_permlane32_swap, kept for A/B runs)
                sc.factor(hi, pm, pe);
                E16_STAMP(4)
                sc.store(lrow, hi, wave);
#else
                sc.factor_lds(hi, Gs + i * GS + 16 * wave, lrow, wave, pm, pe);
                E16_STAMP(4)
#endif
            }
            __syncthreads();
            E16_STAMP(5)
#pragma unroll
            for (int a = 0; a < K; ++a) z[a] = b1[a];
            static_assert(NF == 4 && (K + 1) / 2 <= 8, "pair w and its mirror 7 - w per wave");
            solve_lds<K>(lrow, z, quad, zz);
            static_for<NF>([&](auto w_tag) {
                constexpr int w = decltype(w_tag)::value;
                if (wave == w) {
                    static_for<2>([&](auto i_tag) {
                        constexpr int pp = decltype(i_tag)::value == 0 ? first_pair(w) : second_pair(K, w), c0 = 2 * pp;
                        if constexpr (pp >= 0) {
                            const double zc = (hi && c0 + 1 < K) ? z[c0 + 1 < K ? c0 + 1 : c0] : z[c0];
                            // P = z z^T + Sigma, Sigma = sigma^2 M^-1 (ppca_model.rs:437-439), weighted
                            trpart += minv_pair_lds<K, c0>(lrow, hi, [&](int t, double v) {
                                const bool ok = t > c0 || hi == 0;
                                if (ok && c0 + hi < K) wrow[tri(t, c0) + hi] = wgt * (z[t] * zc + s2 * v);
                            });
                        }
                    });
                }
            });
            } else {
                // k <= 13: the packed factor fits a lane's registers (Posterior<K>, ppca_small.hpp) -- every wave factors every
                // sample (lanes 32-63 mirror 0-31) and the waves share the columns of M^-1, two per instruction stream
            Posterior<K> post;
            post.load([&](int e) { return g0[e]; }, s2);
#pragma unroll
            for (int a = 0; a < K; ++a) z[a] = g0[16 * NTP + a] + b1[a];
            // the W rows go where [G | b] is: every wave holds its operands before any of them writes a row
            __syncthreads();
            post.factor_loaded(pm, pe);
            post.solve_loaded(z, quad, zz);
#pragma unroll
            for (int pp = 0; pp < (K + 1) / 2; ++pp) {
                if (pair_owner(K, pp, NF) != wave) continue;
                const int c0 = 2 * pp;
                const double zc = (hi && c0 + 1 < K) ? z[c0 + 1 < K ? c0 + 1 : c0] : z[c0];
                // P = z z^T + Sigma, Sigma = sigma^2 M^-1 (ppca_model.rs:437-439), weighted
                trpart += post.minv_column_pair(c0, hi, [&](int t, double v, bool ok) {
                    if (ok && c0 + hi < K) wrow[tri(t, c0) + hi] = wgt * (z[t] * zc + s2 * v);
                });
            }
            }
            E16_STAMP(6)
            // tr(C_o Sigma C_o^T) = <Sigma, G> = s2 (K - s2 tr M^-1)  (:345); all-masked samples are filtered out (:333)
            if (m > 0) sc_sq -= wgt * s2 * s2 * trpart;
            if (wave == 0 && hi == 0) {
                const double run_dev = scl[L_DEV + i], run_llk = scl[L_LLK + i], run_w = scl[L_W + i], run_ne = scl[L_NE + i];
                const double run_pm = scl[L_PM + i], run_px = scl[L_PX + i];
                double *zrow = wrow + KP;  // W row = [w P (K') | w z (K) | w | ..]: the layout of the hand-over buffer
#pragma unroll
                for (int a = 0; a < K; ++a) zrow[a] = wgt * z[a];
                zrow[K] = wgt;
                if (m > 0) {
                    sc_sq += wgt * s2 * (double)K;
                    sc_dev += wgt * (0.0 - quad - s2 * zz);  // |x~ - C_o z|^2 minus |x~|^2, added in the epilogue (:346)
                    sc_ne += (row < n) ? 1.0 : 0.0;
                }
                const double lk0 = sample_llk_nolog(0.0, quad, inv_s2, lnsig, m, K);
                if constexpr (WEIGHTED) {
                    if (!p.no_llk) sc_llk += wgt * (m > 0 ? lk0 - 0.5 * Posterior<K>::logdet(pm, pe) : 0.0);
                } else {
                    const bool use = m > 0 && row < n;  // wgt is 1 for real rows
                    sc_llk += use ? lk0 : 0.0;
                    int e;
                    scl[L_PM + i] = frexp(run_pm * (use ? pm : 1.0), &e);
                    scl[L_PX + i] = run_px + (double)(e + (use ? pe : 0));
                }
                sc_w += wgt;
                scl[L_DEV + i] = run_dev + sc_dev;
                scl[L_LLK + i] = run_llk + sc_llk;
                scl[L_W + i] = run_w + sc_w;
                scl[L_NE + i] = run_ne + sc_ne;
            }
            scl[wave * SQW + lane] += sc_sq;
        }
        __syncthreads();  // the tile's W rows are final
        E16_STAMP(7)
        // ------------------------------------------------------------ P4a: cross += X~^T [wz]; the rows leave for HBM
        // behind the MFMAs: 16-byte pieces of the compact rows (column c of a row: wP for c < K', then wz, w, zeros)
        {
            double bzb[2][NCG], axb[2][RT];
            p.Mb[tile * 256 + 64 * wave + lane] = Mb[64 * wave + lane];  // (ahead of the row loads: nothing here may wait for them)
            const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile + 1);
            const int j4 = lane & 3;
#pragma unroll
            for (int c = 0; c < NCG; ++c) bzb[0][c] = Ws[l4 * WS + KP + 4 * c + j4];  // (columns past K of the tile: stale values, in output columns nobody stores)
#pragma unroll
            for (int r = 0; r < RT; ++r) axb[0][r] = Xs[l4 * XS + DW * wave + 16 * r + l15];
            const int64_t row0 = tile * B;
            const int valid = (int)(n - row0 < B ? n - row0 : B);
            double *wout = p.Wrows + row0 * NCOL;
            constexpr int NCH = B * NCOL / 2, CPT = (NCH + 255) / 256;
            // the pieces are read up front, unconditionally, one 16-byte LDS read each (8-byte reads at a 16-byte lane
            // stride are four-way bank conflicts, times four waves: 3.5 k cycles per tile)
            typedef double d2_t __attribute__((ext_vector_type(2)));
            d2_t pv[CPT];
            const int ci0 = 64 * wave + lane;  // (from the opaque lane: nothing of this is hoisted out of the tile loop and parked)
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                int ci = ci0 + 256 * u;
                ci = (NCH % 256 == 0 || ci < NCH) ? ci : NCH - 1;
                const int r = ci / (NCOL / 2), c = 2 * (ci - r * (NCOL / 2));
                const d2_t a = *reinterpret_cast<const d2_t *>(Ws + r * WS + c);
                pv[u] = d2_t{c < NC ? a[0] : 0.0, c + 1 < NC ? a[1] : 0.0};
            }
            auto copy_piece = [&](int u) {
                const int ci = ci0 + 256 * u;
                const int r = ci / (NCOL / 2);  // (past the tile for the pieces of the last round that do not exist)
                if (r < valid) reinterpret_cast<d2_t *>(wout)[ci] = pv[u];  // (non-temporal stores: no difference)
            };
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < RPW) load_row(trs, s);  // unconditional (rows past the end read as zeros)
                if (s + 1 < 8) {
                    const int smp = 4 * (s + 1) + l4;
#pragma unroll
                    for (int c = 0; c < NCG; ++c) bzb[(s + 1) & 1][c] = Ws[smp * WS + KP + 4 * c + j4];
#pragma unroll
                    for (int r = 0; r < RT; ++r) axb[(s + 1) & 1][r] = Xs[smp * XS + DW * wave + 16 * r + l15];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < NCG; ++c)
                        accX[r][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(axb[s & 1][r], bzb[s & 1][c], accX[r][c], 0, 0, 0);
#pragma unroll
                for (int u = s; u < CPT; u += 8) copy_piece(u);
                __builtin_amdgcn_sched_barrier(0);
            }
            E16_STAMP(8)
        }
        __syncthreads();  // the x~ tile, the rows and the sample masks are free
        E16_STAMP(9)
        // ------------------------------------------------------------ P1 of the next tile
        if constexpr (SPLIT) load_cf(lane);
        stage_tile(tile + 1, lane);
        __syncthreads();
        E16_STAMP(10)
    }
#ifdef PPCA_PHASE_TIMING
    if (p.dbg && tid == 0)
        for (int i = 0; i < 11; ++i) p.dbg[(int64_t)blockIdx.x * 16 + i] = (double)tph[i];
#endif

    // ---------------------------------------------------------------- epilogue
    {
        const int lane = lane_entry, l15 = lane & 15, l4 = lane >> 4;
        const double sq_w = wave_sum(scl[wave * SQW + lane]);
        const double xx_w = wave_sum(xx_run);
        if (lane == 0) {
            xxs[wave] = sq_w;
            xxs[NF + wave] = xx_w;
        }
        // sumx: the four waves' per-lane partial sums (lane l: dims 128 h + 2 l + e), added in wave order
#pragma unroll
        for (int q = 0; q < 4; ++q) Xs[wave * cfg::DP + 128 * (q >> 1) + 2 * lane + (q & 1)] = sx_run[q];
        __syncthreads();
        StatsLayout L(d, K);
        double *out = p.part + (int64_t)blockIdx.x * L.len;
        if (tid < d) out[L.sumx + tid] = ((Xs[tid] + Xs[cfg::DP + tid]) + Xs[2 * cfg::DP + tid]) + Xs[3 * cfg::DP + tid];
        if (wave == 0) {
            double v0 = 0.0, xx_tot = 0.0;
#pragma unroll
            for (int w = 0; w < NF; ++w) v0 += xxs[w];
#pragma unroll
            for (int w = 0; w < NF; ++w) xx_tot += xxs[NF + w];
            const int li = lane < B ? lane : 0;
            double sc_llk = scl[L_LLK + li];
            if constexpr (!WEIGHTED) sc_llk -= 0.5 * (log(scl[L_PM + li]) + scl[L_PX + li] * LN_2);
            const double v1 = wave_sum(lane < B ? scl[L_DEV + li] : 0.0), v2 = wave_sum(lane < B ? sc_llk : 0.0),
                         v3 = wave_sum(lane < B ? scl[L_W + li] : 0.0), v4 = wave_sum(lane < B ? scl[L_NE + li] : 0.0);
            if (lane == 0) {
                double *sc = out + L.scalars;
                sc[SC_SQERR] = v0;
                sc[SC_DEVSQ] = v1 + xx_tot;
                sc[SC_LLK] = v2 - 0.5 * inv_s2_k * xx_tot;
                sc[SC_SUMW] = v3;
                sc[SC_NONEMPTY] = v4;
                sc[5] = 0.0;
                sc[6] = 0.0;
                sc[7] = 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int dim = DW * wave + 16 * r + 4 * ((lane >> 2) & 3) + l4;  // D[i][j] of block b: lane 16 i + 4 b + j
            if (dim >= d) continue;
#pragma unroll
            for (int c = 0; c < NCG; ++c) {
                const int col = 4 * c + (lane & 3);
                if (col < K) out[L.cross + (int64_t)dim * K + col] = accX[r][c];
            }
        }
    }
}

// ===================================================================== mask-side statistics
// Diagnostic counters as in ppca_em8.hip: [0] rescales beyond a workgroup's first, [1] periodic flushes, [2] the largest
// number of tiles one workgroup walked, [3] workgroups whose statistics were recomputed in fp64 (rounding check).
__device__ unsigned long long e16_counters[4];

template <int K>
__global__ __launch_bounds__(512) void sstat16_kernel(S16Args p) {
    using cfg = Cfg16<K>;
    constexpr int KP = cfg::KP, B = cfg::B, NC = cfg::NC, NCT = cfg::NCT, NCOL = cfg::NCOL;
    constexpr int NW = 8;
    constexpr int RT = 16 / NW;        // 16-dim row tiles per wave
    constexpr int DW = cfg::DP / NW;   // dims owned by a wave
    constexpr int QW = E16_QW;
    constexpr int IT = 2 * NCOL / NW;  // (column, chunk) items digitised per wave
    constexpr int TILE_CHUNKS = B * NCOL / 2;              // 16-byte pieces of one tile of rows
    constexpr int LDN = (TILE_CHUNKS + 511) / 512;         // ... per thread
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Rs = sm + cfg::S_OFF_R;
    unsigned *Mb = reinterpret_cast<unsigned *>(sm + cfg::S_OFF_MB);
    int *Ex = reinterpret_cast<int *>(sm + cfg::S_OFF_E);
    unsigned *vstamp = reinterpret_cast<unsigned *>(sm + cfg::S_OFF_BAR);
    unsigned char *smb = reinterpret_cast<unsigned char *>(sm);
    constexpr int P0_BYTES = cfg::S_OFF_P0 * 8, PG_BYTES = cfg::S_OFF_P1 * 8;

    const int tid = threadIdx.x, lane_entry = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = p.d;
    const int64_t n = p.n;
    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;

    for (int idx = tid; idx < cfg::DP * 4; idx += 512) Mb[idx] = 0u;
    if (tid == 0) *vstamp = 0u;

    long long accM[RT][NCT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) accM[r][t][q] = 0ll;
    unsigned attempt = 0u;
    int pending = 0, have_scale = 0, flushed = 0, groups = 0;
    int n_rescale = 0, n_flush = 0;  // (wave-uniform: diagnostic counters)
    int rows_win = 0;                // rows cut since the last flush (the window the rounding check of emit() speaks about)
    unsigned *wbad = vstamp + 1;     // set when a flush window's sums are not large against the rounding of the cut
    if (tid == 0) *wbad = 0u;
    StatsLayout L(d, K);
    double *out = p.part + (int64_t)blockIdx.x * L.len;

    // rows of tile t: one buffer descriptor over the rows that exist (the rest read as zeros)
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    u4_t ld[LDN];
    unsigned mbw = 0u;
    auto fetch = [&](int64_t t) {
        int64_t left = n - t * B;
        left = left < 0 ? 0 : (left > B ? B : left);
        const int bytes = __builtin_amdgcn_readfirstlane(t < tile_end ? (int)left * NCOL * 8 : 0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(p.Wrows + (t < tile_end ? t : tile_begin) * B * NCOL), 0, bytes, 0x00020000);
#pragma unroll
        for (int u = 0; u < LDN; ++u) {
            const int ci = tid + 512 * u;
            ld[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, ci < TILE_CHUNKS ? ci * 16 : 0x7FFFFFF0, 0, 0);
        }
        mbw = (t < tile_end && tid < 256) ? p.Mb[t * 256 + tid] : 0u;
    };
    auto park = [&](int64_t t) {  // the fetched tile -> LDS (rows buffer of its parity, mask slot t % 3 of the workgroup's count)
        const int rel = (int)(t - tile_begin);
        u4_t *dst = reinterpret_cast<u4_t *>(Rs + (rel & 1) * B * NCOL);
#pragma unroll
        for (int u = 0; u < LDN; ++u) {
            const int ci = tid + 512 * u;
            if (ci < TILE_CHUNKS) dst[ci] = ld[u];
        }
        if (tid < 256) Mb[tid * 4 + rel % 3] = mbw;
    };

    // ---- digit planes of the tile's rows under the exponents Ex; returns (wave-uniform) whether an entry of a live
    // column did not fit.  Item = (column c, 16-sample chunk): IT items per wave.
    auto digitise = [&](int lane, const double *rows, int dst_bytes) -> bool {
        asm volatile("" : "+v"(lane));
        const bool active = lane < IT;
        const int it = IT * wave + (active ? lane : 0);
        const int c = it >> 1, chunk = it & 1;
        const bool cvalid = c < NC;
        const int E = Ex[c];
        const bool poisoned = E > 5000;
        const double qsc = __hiloint2double((1023 + E16_F - (poisoned ? 0 : E)) << 20, 0);
        const double magic = __hiloint2double(0x43388080, (int)0x80808080);
        unsigned bad = 0u;
        unsigned pl[QW][4];
        const double *wsrcp = rows + (16 * chunk) * NCOL + c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            unsigned wlo[4], whi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // mantissa of w 2^(F - E) + magic = 2^51 + 0x808080808080 + I: xor with the constant's own bits leaves
                // bytes 0..5 = the balanced digits and bits 48..51 = the top digit (4-bit two's complement)
                const double wv = wsrcp[(4 * g4 + j) * NCOL];
                const double v = __builtin_fma(cvalid ? wv : 0.0, qsc, magic);
                const unsigned lo = (unsigned)__double2loint(v) ^ 0x80808080u;
                const unsigned hi = (unsigned)__double2hiint(v) ^ 0x43388080u;
                bad |= hi;  // any exponent field other than 0x433: the entry does not fit (or is not finite)
                const int top = __builtin_amdgcn_sbfe((int)hi, 16, 4);
                wlo[j] = lo;
                whi[j] = __builtin_amdgcn_perm((unsigned)top, hi, 0x0C040100u);  // [d4, d5, d6, 0]
            }
            auto tr4 = [&](const unsigned *w, unsigned *o0, unsigned *o1, unsigned *o2, unsigned *o3) {
                const unsigned t0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), t1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
                const unsigned u0 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), u1 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
                *o0 = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
                *o1 = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                *o2 = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
                if (o3) *o3 = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
            };
            tr4(wlo, &pl[0][g4], &pl[1][g4], &pl[2][g4], &pl[3][g4]);
            tr4(whi, &pl[4][g4], &pl[5][g4], &pl[6][g4], nullptr);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (active) {
            unsigned char *wq = smb + dst_bytes;
#pragma unroll
            for (int sl = 0; sl < QW; ++sl)
                *reinterpret_cast<i4_t *>(wq + ((sl * 2 + chunk) * NCOL + c) * 16) =
                    i4_t{(int)pl[sl][0], (int)pl[sl][1], (int)pl[sl][2], (int)pl[sl][3]};
        }
        const bool mine = active && cvalid && !poisoned && (bad >> 20) != 0u;
        return __builtin_amdgcn_ballot_w64(mine) != 0ull;
    };

    // ---- new exponents from the current tile's column maxima (cold path)
    auto rescale = [&](int lane, const double *rows) {
        asm volatile("" : "+v"(lane));
        const bool active = lane < IT;
        const int it = IT * wave + (active ? lane : 0);
        const int c = it >> 1, chunk = it & 1;
        const bool cvalid = c < NC;
        double m = 0.0;
        bool fin = true;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double av = __builtin_fabs(cvalid ? rows[(16 * chunk + j) * NCOL + c] : 0.0);
            fin = fin && (av < __builtin_inf());
            m = __builtin_fmax(m, av);
        }
        m = __builtin_fmax(m, dpp_f64<0xB1, 0xF>(m));  // the other chunk of the column sits in the neighbouring lane
        const int finw = __builtin_amdgcn_update_dpp(0, fin ? 1 : 0, 0xB1, 0xF, 0xF, true);
        fin = fin && finw != 0;
        const int Eold = have_scale ? Ex[c] : E16_EMIN;
        int Enew = Eold;
        if (m > 0.0) {
            int e = __builtin_amdgcn_frexp_exp(m) + E16_HEAD;  // |w| < 2^(e - HEAD)
            e = e < E16_EMIN ? E16_EMIN : e;
            Enew = e > Eold ? e : Eold;
        }
        if (!fin || Enew > E16_EMAX) Enew = E16_POISON;
        if (Eold > 5000) Enew = Eold;  // (a poisoned column stays poisoned)
        if (active && chunk == 0) Ex[c] = Enew;
    };

    // ---- accumulators -> the workgroup's partial (x 2^(E - F)); C/D row of v_mfma_i32_16x16x64_i8 = 4 (lane / 16) + reg
    auto emit = [&](int lane, bool accumulate, bool clear) {
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
        // The rounding check of this flush window (the guard of the fixed-point form; em8_kernel's is global, behind
        // wguard_kernel -- here it is per workgroup and window, conservative): a row far above its neighbours lifts the column
        // exponents, and a dimension masked in that row sums rows cut far below their resolution.  A diagonal column of S
        // (non-negative terms) of a dimension observed in the window must hold at least 2^34 x the rounding bound
        // 4 sqrt(rows) quanta; otherwise the workgroup's statistics are recomputed in fp64 by sstat16_fallback_kernel.
        if (have_scale && rows_win > 0) {
            constexpr int tW = (KP + K) >> 4, lW = (KP + K) & 15;  // where the window's totals (column w) sit
            const double thr = __builtin_sqrt((double)rows_win) * 0x1p36;
            bool bad = false;
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long seen = __builtin_amdgcn_ballot_w64(accM[r][tW][q] != 0ll);
                    const bool observed = (seen >> ((lane & 48) | lW)) & 1ull;
#pragma unroll
                    for (int t = 0; t < NCT; ++t) {
                        const int c = 16 * t + l15;
                        bool diag = false;
#pragma unroll
                        for (int a = 0; a < K; ++a) diag = diag || (c == tri(a, a));
                        const double mag = __builtin_fabs((double)accM[r][t][q]);
                        bad = bad || (diag && observed && Ex[c] <= 5000 && mag < thr);
                    }
                }
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) *wbad = 1u;
        }
        rows_win = 0;
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const int c = 16 * t + l15, a = c - KP;
            const int E = have_scale ? Ex[c] : 0;
            const double fsc = __hiloint2double(E > 5000 ? 0x7FF80000 : (1023 + E - E16_F) << 20, 0);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int dim = DW * wave + 16 * r + 4 * l4 + q;
                    double v = have_scale ? (double)accM[r][t][q] * fsc : 0.0;
                    if (E > 5000) v = fsc;  // poisoned column: NaN whatever the integers hold
                    if (clear) accM[r][t][q] = 0ll;
                    if (dim < d && c < NC) {
                        double *dst = c < KP ? out + L.S + (int64_t)dim * KP + c
                                             : (a < K ? out + L.U + (int64_t)dim * K + a : out + L.totals + dim);
                        *dst = accumulate ? *dst + v : v;
                    }
                }
        }
    };

    // ---- one contraction: the group in [P0 | P1] (both == false: P0 alone), sample masks of slots slot_first /
    // slot_second; digit sums folded into the int64 accumulators (as em8_kernel's)
    auto contract = [&](int lane, bool both, int slot_first, int slot_second) {
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4, lh = l4 >> 1;
        i4_t af[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const unsigned word = Mb[(DW * wave + 16 * r + l15) * 4 + (lh ? slot_second : slot_first)];
            unsigned f = (word >> (16 * (l4 & 1))) & 0xFFFFu;
            f = (both || lh == 0) ? f : 0u;
            const unsigned g = __builtin_bitreverse32(f);
            af[r][0] = (int)((((g >> 24) & 0xFu) * 0x00204081u) & 0x01010101u);
            af[r][1] = (int)((((g >> 28) & 0xFu) * 0x00204081u) & 0x01010101u);
            af[r][2] = (int)((((g >> 16) & 0xFu) * 0x00204081u) & 0x01010101u);
            af[r][3] = (int)((((g >> 20) & 0xFu) * 0x00204081u) & 0x01010101u);
        }
        const unsigned char *wq = smb + (lh == 0 ? P0_BYTES : PG_BYTES) + ((l4 & 1) * NCOL + l15) * 16;
        constexpr int PSTRIDE = 2 * NCOL * 16;  // bytes between digit planes
        constexpr int NBLK = NCT * 3 * RT;
        i4_t dd[2][3];
        i4_t bb[3];
        auto issue = [&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value, t = i / (3 * RT), batch = (i / RT) % 3, r = i % RT;
            const unsigned char *wt = wq + t * 256 + 3 * batch * PSTRIDE;
            if constexpr (r == 0) {  // the batch's B operands, shared by its row tiles
                bb[0] = *reinterpret_cast<const i4_t *>(wt);
                if constexpr (batch < 2) {
                    bb[1] = *reinterpret_cast<const i4_t *>(wt + PSTRIDE);
                    bb[2] = *reinterpret_cast<const i4_t *>(wt + 2 * PSTRIDE);
                }
            }
            if constexpr (batch < 2) mfma_i8_x3(af[r], bb[0], bb[1], bb[2], dd[i & 1][0], dd[i & 1][1], dd[i & 1][2]);
            else mfma_i8_x1(af[r], bb[0], dd[i & 1][0]);
        };
        auto fold = [&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value, t = i / (3 * RT), batch = (i / RT) % 3, r = i % RT;
            const i4_t *d3 = dd[i & 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // D row = 4 l4 + q (dim 16 r + 4 l4 + q), column l15
                if constexpr (batch == 0) accM[r][t][q] += (long long)((((d3[2][q] << 8) + d3[1][q]) << 8) + d3[0][q]);  // |.| < 2^30
                else if constexpr (batch == 1) accM[r][t][q] += (long long)((((d3[2][q] << 8) + d3[1][q]) << 8) + d3[0][q]) << 24;
                else accM[r][t][q] += (long long)d3[0][q] << 48;
            }
        };
        issue(std::integral_constant<int, 0>{});
        static_for<NBLK>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            if constexpr (i + 1 < NBLK) issue(std::integral_constant<int, i + 1>{});
            fold(i_tag);
        });
    };

    if (tile_begin < tile_end) {
        fetch(tile_begin);
        __syncthreads();  // (the zeroing of Mb)
        park(tile_begin);
    }
    __syncthreads();

    for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int rel = (int)(tile - tile_begin);
        const bool last = tile + 1 == tile_end;
        const int slot_cur = rel % 3, slot_prev = (rel + 2) % 3;
        const double *rows = Rs + (rel & 1) * B * NCOL;
        fetch(tile + 1);  // in flight behind the cut and the contraction
#pragma unroll 1
        for (;;) {  // normally one trip
            ++attempt;
            const bool bad = !have_scale || digitise(lane, rows, pending ? PG_BYTES : P0_BYTES);
            if (bad && lane == 0) __hip_atomic_store(vstamp, attempt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            const bool viol = __builtin_amdgcn_readfirstlane(__hip_atomic_load(vstamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == attempt;
            // a fitting tile completes its group (or is the last tile: alone); a tile that does not fit sends what is
            // pending in alone, under the old exponents -- then (cold path) the integers leave for the partial, the
            // exponents rise and the tile is cut again
            const bool con = pending || (!viol && last);
            if (con) {
                contract(lane, !viol && pending, pending ? slot_prev : slot_cur, slot_cur);
                ++groups;
            }
            pending = (!viol && !con) ? 1 : 0;
            if (!viol) rows_win += B;
            if (viol ? have_scale != 0 : groups >= E16_FLUSH_GROUPS) {
                emit(lane, flushed != 0, true);
                flushed = 1;
                groups = 0;
                n_flush += viol ? 0 : 1;
            }
            if (!viol) break;
            n_rescale += have_scale;
            __syncthreads();  // every wave has read the old exponents
            rescale(lane, rows);
            have_scale = 1;
            __syncthreads();
        }
        park(tile + 1);
        __syncthreads();
    }
    emit(lane_entry, flushed != 0, false);
    __syncthreads();
    if (tid == 0) {
        out[L.scalars + 7] = *wbad ? 1.0 : 0.0;  // read and cleared by sstat16_fallback_kernel (the next launch)
        if (*wbad) atomicAdd(&e16_counters[3], 1ull);
        if (n_rescale) atomicAdd(&e16_counters[0], (unsigned long long)n_rescale);
        if (n_flush) atomicAdd(&e16_counters[1], (unsigned long long)n_flush);
        atomicMax(&e16_counters[2], (unsigned long long)(tile_end > tile_begin ? tile_end - tile_begin : 0));
    }
}

// The fp64 form of sstat16_kernel for the workgroups whose rounding check failed (cold path): S, U, totals of the
// workgroup's tiles summed sample by sample in fp64 from the handed-over rows and the per-tile sample masks, as the
// reference sums them (ppca_model.rs:297-306, :338-348).  Thread = (dimension, column parity); a tile's rows go through
// LDS.  Every workgroup reads its flag (slot 7 of its partial's scalars) and clears it; unflagged ones return at once.
template <int K>
__global__ __launch_bounds__(256) void sstat16_fallback_kernel(S16Args p) {
    using cfg = Cfg16<K>;
    constexpr int KP = cfg::KP, B = cfg::B, NC = cfg::NC, NCOL = cfg::NCOL;
    __shared__ double rows[B * NCOL];
    __shared__ int flag;
    const int tid = threadIdx.x, d = p.d;
    StatsLayout L(d, K);
    double *out = p.part + (int64_t)blockIdx.x * L.len;
    if (tid == 0) {
        flag = out[L.scalars + 7] != 0.0;
        out[L.scalars + 7] = 0.0;
    }
    __syncthreads();
    if (!flag) return;
    const int64_t n = p.n;
    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    const int dim = tid;  // thread = dimension (one wave per SIMD); the columns in two sweeps over the tiles, NU running sums each
    constexpr int NU = (NC + 1) / 2;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        double acc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = 0.0;
        for (int64_t t = tile_begin; t < tile_end; ++t) {
            const int64_t left = n - t * B;
            const int nr = (int)(left < B ? left : B);
            __syncthreads();
            for (int idx = tid; idx < nr * NCOL; idx += 256) rows[idx] = p.Wrows[t * B * NCOL + idx];
            __syncthreads();
            const unsigned mb = p.Mb[t * 256 + dim];  // byte w = rows 8 w .. 8 w + 7 of the tile, row r at bit 7 - r
#pragma unroll 1
            for (int i = 0; i < nr; ++i) {
                const bool on = (mb >> (8 * (i >> 3) + 7 - (i & 7))) & 1u;
                const double *ri = rows + i * NCOL + half * NU;  // (NCOL >= 2 NU: columns past NC are zeros of the hand-over rows)
                static_for<(NU + 15) / 16>([&](auto blk_tag) {  // (sixteen columns at a time: the fence keeps the loads of a row from
                    constexpr int u0 = 16 * decltype(blk_tag)::value;  //  all being hoisted into registers at once)
#pragma unroll
                    for (int u = u0; u < (u0 + 16 < NU ? u0 + 16 : NU); ++u) acc[u] += on ? ri[u] : 0.0;  // select, never multiply
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        if (dim < d) {
#pragma unroll 1
            for (int u = 0; u < NU; ++u) {
                const int c = half * NU + u, a = c - KP;
                if (c >= NC) break;
                double *dst = c < KP ? out + L.S + (int64_t)dim * KP + c : (a < K ? out + L.U + (int64_t)dim * K + a : out + L.totals + dim);
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < NU; ++w) v = (w == u) ? acc[w] : v;  // (register arrays are indexed by compile-time constants only)
                *dst = v;
            }
        }
    }
}

hipError_t em16_debug_counters(unsigned long long *out4, int reset, hipStream_t s) {
    if (hipError_t e = hipMemcpyFromSymbolAsync(out4, HIP_SYMBOL(e16_counters), sizeof(unsigned long long) * 4, 0, hipMemcpyDeviceToHost, s);
        e != hipSuccess)
        return e;
    if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) return e;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(e16_counters), z, sizeof(z), 0, hipMemcpyHostToDevice, s); e != hipSuccess) return e;
        return hipStreamSynchronize(s);
    }
    return hipSuccess;
}

// ------------------------------------------------------------------ launchers
template <class Kern>
static hipError_t set_lds_once(Kern kern, size_t lds, std::atomic<unsigned long long> &done) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

template <int K>
static hipError_t launch_em16_t(int grid, const Em16Launch &a, hipStream_t s) {
    using cfg = Cfg16<K>;
    // slice table + guard flags of the model (device-side; the fp64 Gram rows of a model that trips the guard come from
    // the caller's guarded launch, see ppca_generic.hip)
    const Em16Launch &e = a;
    const size_t lds_e = sizeof(double) * cfg::LDS_DOUBLES, lds_s = sizeof(double) * cfg::S_LDS_DOUBLES;
    static std::atomic<unsigned long long> done_w{0ull}, done_u{0ull}, done_s{0ull};
    if (a.w) {
        if (hipError_t er = set_lds_once(&estep16_kernel<K, true>, lds_e, done_w); er != hipSuccess) return er;
        hipLaunchKernelGGL((estep16_kernel<K, true>), dim3(grid), dim3(256), lds_e, s, e);
    } else {
        if (hipError_t er = set_lds_once(&estep16_kernel<K, false>, lds_e, done_u); er != hipSuccess) return er;
        hipLaunchKernelGGL((estep16_kernel<K, false>), dim3(grid), dim3(256), lds_e, s, e);
    }
    if (hipError_t er = hipGetLastError(); er != hipSuccess) return er;
    S16Args b{};
    b.Wrows = a.Wrows; b.Mb = a.Mb; b.n = a.n; b.d = a.d; b.part = a.part;
    if (hipError_t er = set_lds_once(&sstat16_kernel<K>, lds_s, done_s); er != hipSuccess) return er;
    hipLaunchKernelGGL((sstat16_kernel<K>), dim3(grid), dim3(512), lds_s, s, b);
    if (hipError_t er = hipGetLastError(); er != hipSuccess) return er;
    hipLaunchKernelGGL((sstat16_fallback_kernel<K>), dim3(grid), dim3(256), 0, s, b);  // (returns at once unless the check failed)
    return hipGetLastError();
}

bool em16_covers(int d, int k) { return d >= 1 && d <= 256 && k >= E16_MIN_K && k <= E16_MAX_K; }
int em16_ncol(int k) { return ((k * (k + 1) / 2 + k + 1 + 15) / 16) * 16; }
size_t em16_qtab_bytes(int k) { return (size_t)((k * (k + 1) / 2 + 15) / 16) * QS * 4 * 1024; }

#ifdef PPCA_E16_ONLY  // kernel-tuning builds: one instantiation
#define PPCA_E16_ALL(M) M(PPCA_E16_ONLY)
#else
#define PPCA_E16_ALL(M) M(11) M(12) M(13) M(14) M(15) M(16)
#endif

hipError_t launch_em16(int k, int grid, const Em16Launch &a, hipStream_t s) {
    switch (k) {
#define PPCA_E16_CASE(KK) \
    case KK:              \
        return launch_em16_t<KK>(grid, a, s);
        PPCA_E16_ALL(PPCA_E16_CASE)
#undef PPCA_E16_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ppca
