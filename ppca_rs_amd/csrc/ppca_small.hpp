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
    __builtin_amdgcn_sched_barrier(0);  // measured: fences cost 9 % of P3 and save no registers
#endif
}

PPCA_HD double fast_rsqrt(double s) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rsq_f64 seed + one third-order correction: y (1 + e/2 + 3 e^2/8), e = 1 - s y^2 (residual ~ e^3).
    // Pivots of M = sigma^2 I + G are positive normal numbers; the library routine's scaling for
    // denormals / huge arguments is not needed and sat on the critical path of every column.
    const double y = __builtin_amdgcn_rsq(s);
    const double e = fma(-s * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
#else
    return 1.0 / std::sqrt(s);
#endif
}

// ---------------------------------------------------------------------------
// Posterior of one sample, compile-time K, packed Cholesky factor in registers.
//   factor(gload, s2)    M = G + s2 I = L L^T, gload(e) -> packed Gram entry e; returns ln det M
//                        (diagonal slots keep 1/L_aa, the only form ever used)
//   solve(bload, z, ..)  z = M^-1 b, quad = b^T M^-1 b, zz = |z|^2
//   minv_column(c, st)   column c of M^-1 (rows a >= c) by two triangular solves, st(a, c, v);
//                        returns (M^-1)_cc.  Columns are independent, so the waves of a
//                        workgroup share them (column_owner) after factoring redundantly.
template <int K>
struct Posterior {
    static constexpr int KP = K * (K + 1) / 2;
    double L[KP];

    // det M = pm * 2^pe with pm in [0.25, 1): the pivots are multiplied in two groups, each split by one
    // frexp (safe unless five pivots alone overflow a double; the reference takes ln of the whole LU
    // determinant, output_covariance.rs:115-121).  The caller decides when to take the logarithm.
    // Right-looking (outer-product) Cholesky: every column step is a set of INDEPENDENT multiply-adds on
    // the trailing packed matrix.  The row-by-row form leaves each entry as one chain of dependent FMAs and
    // ran at ~12 cycles per operation on one wave per SIMD instead of the ~6 the fp64 pipe issues at.
    template <class GLoad>
    PPCA_HD void factor(GLoad gload, double s2, double &pm, int &pe) {
        load(gload, s2);
        factor_loaded(pm, pe);
    }
    // factor() in two steps, for callers that must synchronise between reading G and the rest (the Gram is
    // overwritten in place by the results): M = G + s2 I into the registers, then the factorisation proper
    template <class GLoad>
    PPCA_HD void load(GLoad gload, double s2) {
#pragma unroll
        for (int e = 0; e < KP; ++e) L[e] = gload(e);
#pragma unroll
        for (int a = 0; a < K; ++a) L[tri(a, a)] += s2;
    }
    PPCA_HD void factor_loaded(double &pm, int &pe) {
        double grp[2] = {1.0, 1.0};
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const double piv = L[tri(c, c)];
            grp[c >= (K + 1) / 2] *= piv;
            const double inv = fast_rsqrt(piv);  // 1 / L_cc
            L[tri(c, c)] = inv;
#pragma unroll
            for (int a = c + 1; a < K; ++a) L[tri(a, c)] *= inv;
#pragma unroll
            for (int a = c + 1; a < K; ++a)
#pragma unroll
                for (int b = c + 1; b <= a; ++b) L[tri(a, b)] -= L[tri(a, c)] * L[tri(b, c)];
        }
        int e0, e1;
        pm = frexp(grp[0], &e0) * frexp(grp[1], &e1);
        pe = e0 + e1;
    }
    PPCA_HD static double logdet(double pm, int pe) { return log(pm) + (double)pe * LN_2; }

    // Substitutions are column-oriented for the same reason: once an unknown is final, the updates of all
    // the remaining right-hand sides are independent.
    template <class BLoad>
    PPCA_HD void solve(BLoad bload, double (&z)[K], double &quad, double &zz) const {
#pragma unroll
        for (int a = 0; a < K; ++a) z[a] = bload(a);
        solve_loaded(z, quad, zz);
    }
    // the same with b already in z
    PPCA_HD void solve_loaded(double (&z)[K], double &quad, double &zz) const {
        quad = 0.0;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            z[t] *= L[tri(t, t)];  // y = L^-1 b
            quad += z[t] * z[t];
#pragma unroll
            for (int a = t + 1; a < K; ++a) z[a] -= L[tri(a, t)] * z[t];
        }
        zz = 0.0;
#pragma unroll
        for (int t = K - 1; t >= 0; --t) {
            z[t] *= L[tri(t, t)];  // z = L^-T y
            zz += z[t] * z[t];
#pragma unroll
            for (int a = 0; a < t; ++a) z[a] -= L[tri(t, a)] * z[t];
        }
        sched_fence();
    }

    // quad = b^T M^-1 b = |L^-1 b|^2 alone (the log-likelihood needs no z): the forward substitution of solve()
    template <class BLoad>
    PPCA_HD double forward_quad(BLoad bload) const {
        double y[K];
#pragma unroll
        for (int a = 0; a < K; ++a) y[a] = bload(a);
        double quad = 0.0;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            y[t] *= L[tri(t, t)];
            quad += y[t] * y[t];
#pragma unroll
            for (int a = t + 1; a < K; ++a) y[a] -= L[tri(a, t)] * y[t];
        }
        return quad;
    }

    template <class Store>
    PPCA_HD double minv_column(int c, Store st) const {
        double u[K];
#pragma unroll
        for (int a = 0; a < K; ++a) u[a] = (a == c) ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            if (t < c) continue;
            u[t] *= L[tri(t, t)];
#pragma unroll
            for (int a = t + 1; a < K; ++a) u[a] -= L[tri(a, t)] * u[t];
        }
#pragma unroll
        for (int t = K - 1; t >= 0; --t) {
            if (t < c) continue;
            u[t] *= L[tri(t, t)];  // x_t final; rows above c are not needed (symmetry)
            st(t, c, u[t]);
#pragma unroll
            for (int a = 0; a < t; ++a)
                if (a >= c) u[a] -= L[tri(t, a)] * u[t];
        }
        sched_fence();
        return u[c];
    }

    // Two columns at once, one per half of the wave (hi = 0: column c0, hi = 1: column c0 + 1), with the loop
    // bounds of c0: a solver lane pair (lane, lane + 32) holds the same factor, so the half-empty wave of the
    // 32-sample tile does the columns' work two at a time.  st(t, v, ok): entry (t, c0 + hi) = v when ok.
    template <class Store>
    PPCA_HD double minv_column_pair(int c0, int hi, Store st) const {
        double u[K];
#pragma unroll
        for (int a = 0; a < K; ++a) u[a] = (a == c0 + hi) ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            if (t < c0) continue;
            u[t] *= L[tri(t, t)];
#pragma unroll
            for (int a = t + 1; a < K; ++a) u[a] -= L[tri(a, t)] * u[t];
        }
        double diag = 0.0;
#pragma unroll
        for (int t = K - 1; t >= 0; --t) {
            if (t < c0) continue;
            u[t] *= L[tri(t, t)];
            st(t, u[t], t > c0 || hi == 0);
            if (t == c0 + 1) diag = hi ? u[t] : diag;
            if (t == c0) diag = hi ? diag : u[t];
#pragma unroll
            for (int a = 0; a < t; ++a)
                if (a >= c0) u[a] -= L[tri(t, a)] * u[t];
        }
        sched_fence();
        return diag;  // (M^-1) at (c0 + hi, c0 + hi); 0 for the unpaired half of an odd last column
    }
};

// Which of nw workers owns column c of M^-1: greedy balance of the column costs -- (K - c)^2 multiply-adds
// of the two substitutions plus 4 (K - c) for forming and storing the second-moment entries -- with worker
// 0 handicapped by what it alone does (z row, llk, noise terms: ~1.2 x a quarter of all columns at K = 10).
PPCA_HD constexpr int column_owner(int K, int c, int nw) {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int total = 0;
    for (int cc = 0; cc < K; ++cc) total += (K - cc) * (K - cc) + 4 * (K - cc);
    load[0] = total / nw;
    int own = 0;
    for (int cc = 0; cc <= c; ++cc) {
        own = 0;
        for (int w = 1; w < nw; ++w)
            if (load[w] < load[own]) own = w;
        load[own] += (K - cc) * (K - cc) + 4 * (K - cc);
    }
    return own;
}

// The same for column PAIRS (2p, 2p + 1), each costing what its first column costs (Posterior::minv_column_pair).
PPCA_HD constexpr int pair_owner(int K, int p, int nw) {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int total = 0;
    for (int cc = 0; cc < K; ++cc) total += (K - cc) * (K - cc) + 4 * (K - cc);
    load[0] = total / (nw + 1);  // worker 0's own extras, in the same units
    int own = 0;
    for (int pp = 0; pp <= p; ++pp) {
        own = 0;
        for (int w = 1; w < nw; ++w)
            if (load[w] < load[own]) own = w;
        const int c0 = 2 * pp;
        load[own] += (K - c0) * (K - c0) + 4 * (K - c0);
    }
    return own;
}

// The same balance over nw EQUAL workers (ppca_em9.hip: the three front waves that are not the tile's solver).
PPCA_HD constexpr int pair_owner_eq(int K, int p, int nw, int handicap_last = 0) {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load[nw - 1] = handicap_last;  // what the last worker does besides (the per-sample scalars: ~107 of these units)
    int own = 0;
    for (int pp = 0; pp <= p; ++pp) {
        own = 0;
        for (int w = 1; w < nw; ++w)
            if (load[w] < load[own]) own = w;
        const int c0 = 2 * pp;
        load[own] += (K - c0) * (K - c0) + 4 * (K - c0);
    }
    return own;
}
PPCA_HD constexpr int column_owner_eq(int K, int c, int nw, int handicap_last = 0) {
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load[nw - 1] = handicap_last;
    int own = 0;
    for (int cc = 0; cc <= c; ++cc) {
        own = 0;
        for (int w = 1; w < nw; ++w)
            if (load[w] < load[own]) own = w;
        load[own] += (K - cc) * (K - cc) + 4 * (K - cc);
    }
    return own;
}

// true when no earlier pair belongs to pair p's owner (that pair's part of the factor covers the owner's later pairs)
PPCA_HD constexpr bool pair_first_of_owner(int K, int p, int nw) {
    for (int q = 0; q < p; ++q)
        if (pair_owner(K, q, nw) == pair_owner(K, p, nw)) return false;
    return true;
}

// Per-sample log-likelihood from the solve's by-products (ppca_model.rs:124-139):
//   -1/2 [ (|x~|^2 - b^T M^-1 b)/s2 + ln det M + 2 ln(sigma)(m - k) + ln(2 pi) m ],  0 if m == 0
PPCA_HD double sample_llk(double xx, double quad, double logdet, double s2, double ln_sigma, int m, int k) {
    if (m == 0) return 0.0;
    return -0.5 * ((xx - quad) / s2 + logdet + 2.0 * ln_sigma * (double)(m - k) + LN_2PI * (double)m);
}
// The same without the ln det M term and with 1/s2 precomputed (the fused EM pass adds the logarithms
// of a whole run of samples at once).
PPCA_HD double sample_llk_nolog(double xx, double quad, double inv_s2, double ln_sigma, int m, int k) {
    if (m == 0) return 0.0;
    return -0.5 * ((xx - quad) * inv_s2 + 2.0 * ln_sigma * (double)(m - k) + LN_2PI * (double)m);
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
