// ppca_small.hpp -- small dense SPD routines shared by the HIP kernels and the
// host-side finalisation.  Symmetric k x k matrices are lower-packed:
// entry (a, b), a >= b, lives at tri(a, b) = a(a+1)/2 + b.
//
// What this restates (reference = viodotcom/ppca_rs, paths relative to its root):
//   posterior of one sample        ppca/src/ppca_model.rs:195-208 via
//                                  ppca/src/output_covariance.rs:57-101
//   log-likelihood of one sample   ppca/src/ppca_model.rs:124-139 via
//                                  ppca/src/output_covariance.rs:115-142
//   row systems of the M-step      ppca/src/ppca_model.rs:309-322
// in the numerically stable form  z = M^-1 b,  Sigma = sigma^2 M^-1,
// ln det M = sum ln(pivots)  (the reference uses the subtractive Woodbury form and
// LU determinant; tests bound the difference against the literal oracle).
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define PPCA_HD __host__ __device__ __forceinline__
#else
#define PPCA_HD inline
#endif

namespace ppca {

constexpr double LN_2PI = 1.8378770664093453;  // ppca_model.rs:16
constexpr double LN_2 = 0.6931471805599453;

PPCA_HD constexpr int tri(int a, int b) { return a * (a + 1) / 2 + b; }

// Packed statistic buffer layout (see include/ppca_hip.h).
struct StatsLayout {
    int64_t cross, S, U, sumx, totals, scalars, len;
    int d, k, kp;
    PPCA_HD StatsLayout(int d_, int k_) : d(d_), k(k_), kp(k_ * (k_ + 1) / 2) {
        cross = 0;
        S = cross + (int64_t)d * k;
        U = S + (int64_t)d * kp;
        sumx = U + (int64_t)d * k;
        totals = sumx + d;
        scalars = totals + d;
        len = scalars + 8;
    }
};
enum { SC_SQERR = 0, SC_DEVSQ = 1, SC_LLK = 2, SC_SUMW = 3, SC_NONEMPTY = 4 };

// Scheduling fence (device only): keeps hipcc from hoisting a whole phase's LDS
// operand loads to its top, which would double the live registers of the solve.
PPCA_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PPCA_SOLVE_FENCES)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

PPCA_HD double fast_rsqrt(double s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(s);
#else
    return 1.0 / std::sqrt(s);
#endif
}

// ---------------------------------------------------------------------------
// Posterior of one sample, compile-time K.  Only the packed Cholesky factor
// (K' doubles) is live in registers; everything else streams through callbacks
// so that the caller decides where operands and results live (LDS rows):
//   gload(e)        -> packed Gram entry e of G = C_o^T C_o
//   bload(a)        -> entry a of b = C_o^T x~
//   zstore(a, v)    <- z_a, z = M^-1 b with M = G + s2 I
//   mstore(a, c, v) <- (M^-1)_{ac}, a >= c       (Sigma = s2 M^-1)
// and returns quad = b^T M^-1 b, zz = |z|^2, logdet = ln det M, trminv = tr M^-1.
// Phases: left-looking Cholesky (diagonal slots keep 1/L_aa, the only form ever
// used), forward/backward substitution, in-place inverse of the triangular factor,
// M^-1 = L^-T L^-1 entry by entry.
template <int K, class GLoad, class BLoad, class ZStore, class MStore>
PPCA_HD void posterior_solve(GLoad gload, BLoad bload, double s2, ZStore zstore, MStore mstore, double &quad,
                             double &zz, double &logdet, double &trminv) {
    constexpr int KP = K * (K + 1) / 2;
    double L[KP];
    double mant = 1.0;
    int ex = 0;
#pragma unroll
    for (int a = 0; a < K; ++a) {
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            double s = gload(tri(a, c));
            if (c == a) s += s2;
#pragma unroll
            for (int t = 0; t < c; ++t) s -= L[tri(a, t)] * L[tri(c, t)];
            if (c < a) {
                L[tri(a, c)] = s * L[tri(c, c)];
            } else {
                L[tri(a, a)] = fast_rsqrt(s);  // 1 / L_aa
                int e;
                mant *= frexp(s, &e);  // ln det M = ln prod(pivots), overflow-safe
                ex += e;
            }
        }
        sched_fence();
    }
    logdet = log(mant) + (double)ex * LN_2;
    {
        double y[K];
        quad = 0.0;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            double s = bload(a);
#pragma unroll
            for (int t = 0; t < a; ++t) s -= L[tri(a, t)] * y[t];
            y[a] = s * L[tri(a, a)];
            quad += y[a] * y[a];
        }
        zz = 0.0;
#pragma unroll
        for (int a = K - 1; a >= 0; --a) {
            double s = y[a];
#pragma unroll
            for (int t = a + 1; t < K; ++t) s -= L[tri(t, a)] * y[t];
            y[a] = s * L[tri(a, a)];  // z_a overwrites y_a
            zz += y[a] * y[a];
            zstore(a, y[a]);
        }
        sched_fence();
    }
    // in-place inverse of the lower-triangular factor (columns right to left, rows
    // bottom to top, so every original entry is read before it is overwritten)
#pragma unroll
    for (int j = K - 2; j >= 0; --j) {
#pragma unroll
        for (int a = K - 1; a > j; --a) {
            double s = 0.0;
#pragma unroll
            for (int t = j + 1; t <= a; ++t) s += L[tri(a, t)] * L[tri(t, j)];
            L[tri(a, j)] = -s * L[tri(j, j)];
        }
        sched_fence();
    }
    trminv = 0.0;
#pragma unroll
    for (int a = 0; a < K; ++a) {
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            double s = 0.0;
#pragma unroll
            for (int t = a; t < K; ++t) s += L[tri(t, a)] * L[tri(t, c)];
            if (c == a) trminv += s;
            mstore(a, c, s);
        }
        sched_fence();
    }
}

// Per-sample log-likelihood from the solve's by-products (ppca_model.rs:124-139):
//   -1/2 [ (|x~|^2 - b^T M^-1 b)/s2 + ln det M + 2 ln(sigma)(m - k) + ln(2 pi) m ],  0 if m == 0
PPCA_HD double sample_llk(double xx, double quad, double logdet, double s2, double ln_sigma, int m, int k) {
    if (m == 0) return 0.0;
    return -0.5 * ((xx - quad) / s2 + logdet + 2.0 * ln_sigma * (double)(m - k) + LN_2PI * (double)m);
}

// ---------------------------------------------------------------------------
// Run-time-k routines on packed arrays in memory (host finalisation, generic path).

// In-place Cholesky of packed SPD a (k x k).  Returns false when a pivot is not
// strictly positive / not finite (the caller then keeps the old row,
// ppca_model.rs:313-321).
PPCA_HD bool chol_packed(double *a, int k) {
    for (int r = 0; r < k; ++r) {
        for (int c = 0; c <= r; ++c) {
            double s = a[tri(r, c)];
            for (int t = 0; t < c; ++t) s -= a[tri(r, t)] * a[tri(c, t)];
            if (c < r) {
                a[tri(r, c)] = s / a[tri(c, c)];
            } else {
                if (!(s > 0.0) || !(s < 1.0e308)) return false;
                a[tri(r, r)] = sqrt(s);
            }
        }
    }
    return true;
}

// Solve (L L^T) x = rhs in place on x.
PPCA_HD void chol_solve_packed(const double *l, int k, double *x) {
    for (int r = 0; r < k; ++r) {
        double s = x[r];
        for (int t = 0; t < r; ++t) s -= l[tri(r, t)] * x[t];
        x[r] = s / l[tri(r, r)];
    }
    for (int r = k - 1; r >= 0; --r) {
        double s = x[r];
        for (int t = r + 1; t < k; ++t) s -= l[tri(t, r)] * x[t];
        x[r] = s / l[tri(r, r)];
    }
}

// One row of the M-step (ppca_model.rs:297-322), compile-time K, registers only.
// S packed (K'), rhs (K) -> out (K).  Returns false if S + tau I is not SPD.
template <int K>
PPCA_HD bool row_solve(const double *S, double tau, const double *rhs, double *out) {
    constexpr int KP = K * (K + 1) / 2;
    double L[KP];
    double rinv[K];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < K; ++a) {
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            double s = S[tri(a, c)];
            if (c == a) s += tau;
#pragma unroll
            for (int t = 0; t < c; ++t) s -= L[tri(a, t)] * L[tri(c, t)];
            if (c < a) {
                L[tri(a, c)] = s * rinv[c];
            } else {
                ok = ok && (s > 0.0) && (s < 1.0e308);
                double r = 1.0 / sqrt(s);
                rinv[a] = r;
                L[tri(a, a)] = s * r;
            }
        }
    }
    double y[K];
#pragma unroll
    for (int a = 0; a < K; ++a) {
        double s = rhs[a];
#pragma unroll
        for (int t = 0; t < a; ++t) s -= L[tri(a, t)] * y[t];
        y[a] = s * rinv[a];
    }
    double x[K];
#pragma unroll
    for (int a = K - 1; a >= 0; --a) {
        double s = y[a];
#pragma unroll
        for (int t = a + 1; t < K; ++t) s -= L[tri(t, a)] * x[t];
        x[a] = s * rinv[a];
    }
    if (ok) {
#pragma unroll
        for (int a = 0; a < K; ++a) out[a] = x[a];
    }
    return ok;
}

}  // namespace ppca
