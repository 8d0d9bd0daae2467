/*
 * ppca_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A literal C restatement of the reference's rayon/nalgebra PPCA hot path
 * (viodotcom/ppca_rs @ 2024_10_08).  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load this; the shipped HIP path never does.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  The structure is deliberately the reference's own:
 *   - per-sample `infer_one` with the subtractive Woodbury form,
 *   - a materialised list of posteriors (N x (k + k*k) doubles),
 *   - the M-step as FOUR sweeps (cross moment; d sequential scans parallel over
 *     d; noise 4-tuple reduce; mean), exactly as ppca_model.rs:277-393,
 * so that (a) the GPU's fused/stable formulation is checked against the
 * reference's literal arithmetic, not against itself, and (b) this file timed
 * with OpenMP is an honest stand-in for the rayon CPU path (kind = "port").
 *
 * PARITY PINNING.  The reference cannot be built here (Rust; no cargo/rustc) and
 * its own tests hold exactly two known-answer values, both reproduced by
 * tests/test_oracle.py:
 *     quadratic_form      34.219269102989976   (ppca_model.rs:658-665, 6 digits)
 *     covariance_log_det  -3.4932797763741386  (ppca_model.rs:667-671, 6 digits)
 * plus test_llk's inputs (ppca_model.rs:673-680; no expected value upstream).
 * infer_one / iterate_with_prior / smooth / extrapolate / to_canonical / mixture
 * are NOT pinned by any reference test or artefact: for those rows the oracle is
 * "parity unpinned" and is anchored instead by an independent dense-Gaussian
 * evaluation (scipy) in tests/test_oracle.py.
 *
 * Third-party arithmetic not under /root/reference: nalgebra 0.32.2
 * (try_inverse, determinant, qr().solve, svd) restated here from the published
 * algorithms: LU with partial pivoting, Householder QR, one-sided Jacobi SVD.
 *
 * Matrices are row-major unless noted.  All arithmetic is f64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LN_2PI 1.8378770664093453 /* ppca_model.rs:16 */

static int is_finite(double v) { return isfinite(v); }

/* test infrastructure: the number of OpenMP threads the loops below use (many tiny calls on a many-core host --
 * a mixture of hundreds of components over a few hundred samples -- spend their time starting teams); returns the
 * previous value */
int ppca_oracle_set_threads(int n) {
#ifdef _OPENMP
    const int old = omp_get_max_threads();
    if (n >= 1) omp_set_num_threads(n);
    return old;
#else
    (void)n;
    return 1;
#endif
}

int ppca_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- nalgebra */

/* LU with partial pivoting, in place on a (n x n).  Returns sign of the
 * permutation, 0 if a zero pivot is met.  (nalgebra::linalg::LU) */
static int lu_decompose(double *a, int n, int *piv) {
    int sign = 1;
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(a[r * n + c]) > best) { best = fabs(a[r * n + c]); p = r; }
        piv[c] = p;
        if (best == 0.0) return 0;
        if (p != c) {
            for (int j = 0; j < n; ++j) { double t = a[c * n + j]; a[c * n + j] = a[p * n + j]; a[p * n + j] = t; }
            sign = -sign;
        }
        double inv = 1.0 / a[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            double f = a[r * n + c] * inv;
            a[r * n + c] = f;
            for (int j = c + 1; j < n; ++j) a[r * n + j] -= f * a[c * n + j];
        }
    }
    return sign;
}

/* Matrix::try_inverse (output_covariance.rs:68, prior.rs:39).  0 on failure. */
static int mat_inverse(const double *m, int n, double *out) {
    if (n == 0) return 1;
    double *a = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    memcpy(a, m, sizeof(double) * n * n);
    int ok = lu_decompose(a, n, piv) != 0;
    if (ok) {
        for (int col = 0; col < n; ++col) {
            /* solve A x = e_col */
            double *x = out; /* write column col of out (row-major: out[r*n+col]) */
            double rhs[64];
            double *b = n <= 64 ? rhs : (double *)malloc(sizeof(double) * n);
            for (int r = 0; r < n; ++r) b[r] = (r == col) ? 1.0 : 0.0;
            for (int c = 0; c < n; ++c) { int p = piv[c]; if (p != c) { double t = b[c]; b[c] = b[p]; b[p] = t; } }
            for (int r = 0; r < n; ++r) { double s = b[r]; for (int j = 0; j < r; ++j) s -= a[r * n + j] * b[j]; b[r] = s; }
            for (int r = n - 1; r >= 0; --r) { double s = b[r]; for (int j = r + 1; j < n; ++j) s -= a[r * n + j] * b[j]; b[r] = s / a[r * n + r]; }
            for (int r = 0; r < n; ++r) x[r * n + col] = b[r];
            if (b != rhs) free(b);
        }
    }
    free(a); free(piv);
    return ok;
}

/* Matrix::determinant (output_covariance.rs:117) */
static double mat_determinant(const double *m, int n) {
    if (n == 0) return 1.0;
    double *a = (double *)malloc(sizeof(double) * n * n);
    int *piv = (int *)malloc(sizeof(int) * n);
    memcpy(a, m, sizeof(double) * n * n);
    int sign = lu_decompose(a, n, piv);
    double det = (double)sign;
    if (sign != 0) for (int i = 0; i < n; ++i) det *= a[i * n + i];
    free(a); free(piv);
    return det;
}

/* Matrix::qr().solve(b) (ppca_model.rs:310-312, prior.rs:106-108): Householder
 * QR, then back substitution; "None" (return 0) when R has a zero diagonal. */
static int qr_solve(const double *m, int n, const double *rhs, double *x) {
    if (n == 0) return 1;
    double *a = (double *)malloc(sizeof(double) * n * n);
    double *b = (double *)malloc(sizeof(double) * n);
    double *v = (double *)malloc(sizeof(double) * n);
    memcpy(a, m, sizeof(double) * n * n);
    memcpy(b, rhs, sizeof(double) * n);
    int ok = 1;
    for (int c = 0; c < n; ++c) {
        double norm = 0.0;
        for (int r = c; r < n; ++r) norm += a[r * n + c] * a[r * n + c];
        norm = sqrt(norm);
        if (norm == 0.0) { ok = 0; break; }
        double alpha = a[c * n + c] > 0.0 ? -norm : norm;
        double vnorm2 = 0.0;
        for (int r = c; r < n; ++r) { v[r] = a[r * n + c]; }
        v[c] -= alpha;
        for (int r = c; r < n; ++r) vnorm2 += v[r] * v[r];
        if (vnorm2 > 0.0) {
            for (int j = c; j < n; ++j) {
                double dot = 0.0;
                for (int r = c; r < n; ++r) dot += v[r] * a[r * n + j];
                double f = 2.0 * dot / vnorm2;
                for (int r = c; r < n; ++r) a[r * n + j] -= f * v[r];
            }
            double dot = 0.0;
            for (int r = c; r < n; ++r) dot += v[r] * b[r];
            double f = 2.0 * dot / vnorm2;
            for (int r = c; r < n; ++r) b[r] -= f * v[r];
        }
    }
    if (ok) {
        for (int r = n - 1; r >= 0; --r) {
            double s = b[r];
            for (int j = r + 1; j < n; ++j) s -= a[r * n + j] * x[j];
            if (a[r * n + r] == 0.0) { ok = 0; break; }
            x[r] = s / a[r * n + r];
        }
    }
    free(a); free(b); free(v);
    return ok;
}

/* ------------------------------------------------ output_covariance.rs */

/* inner_product :57-59   G = C_o^T C_o   (co is m x k) */
static void oc_inner_product(const double *co, int m, int k, double *g) {
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
            double s = 0.0;
            for (int r = 0; r < m; ++r) s += co[r * k + a] * co[r * k + b];
            g[a * k + b] = s;
        }
}

/* inner_matrix :61-64   I*sigma^2 + G */
static void oc_inner_matrix(const double *co, int m, int k, double sigma, double *mm) {
    oc_inner_product(co, m, k, mm);
    double s2 = sigma * sigma; /* powi(2) */
    for (int a = 0; a < k; ++a) mm[a * k + a] = 1.0 * s2 + mm[a * k + a];
}

/* inner_inverse :66-70 */
static void oc_inner_inverse(const double *co, int m, int k, double sigma, double *inv) {
    double *mm = (double *)malloc(sizeof(double) * (k * k + 1));
    oc_inner_matrix(co, m, k, sigma, mm);
    mat_inverse(mm, k, inv); /* "inner matrix is always invertible" */
    free(mm);
}

/* estimator_transform :90-94   E = (C_o^T - G * Minv * C_o^T) / sigma^2   (k x m) */
static void oc_estimator_transform(const double *co, int m, int k, double sigma, double *e) {
    double *g = (double *)malloc(sizeof(double) * (k * k + 1));
    double *inv = (double *)malloc(sizeof(double) * (k * k + 1));
    double *gi = (double *)malloc(sizeof(double) * (k * k + 1));
    oc_inner_product(co, m, k, g);
    oc_inner_inverse(co, m, k, sigma, inv);
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
            double s = 0.0;
            for (int c = 0; c < k; ++c) s += g[a * k + c] * inv[c * k + b];
            gi[a * k + b] = s;
        }
    double s2 = sigma * sigma;
    for (int a = 0; a < k; ++a)
        for (int r = 0; r < m; ++r) {
            double s = 0.0;
            for (int c = 0; c < k; ++c) s += gi[a * k + c] * co[r * k + c];
            e[a * m + r] = (co[r * k + a] - s) / s2;
        }
    free(g); free(inv); free(gi);
}

/* estimator_covariance :98-101   I - E * C_o  (estimator_transform is evaluated again) */
static void oc_estimator_covariance(const double *co, int m, int k, double sigma, double *cov) {
    double *e = (double *)malloc(sizeof(double) * (k * m + 1));
    oc_estimator_transform(co, m, k, sigma, e);
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
            double s = 0.0;
            for (int r = 0; r < m; ++r) s += e[a * m + r] * co[r * k + b];
            cov[a * k + b] = (a == b ? 1.0 : 0.0) - s;
        }
    free(e);
}

/* covariance_log_det :115-121 */
static double oc_covariance_log_det(const double *co, int m, int k, double sigma) {
    double *mm = (double *)malloc(sizeof(double) * (k * k + 1));
    oc_inner_matrix(co, m, k, sigma, mm);
    double det = mat_determinant(mm, k);
    free(mm);
    return log(det) + log(sigma) * 2.0 * ((double)m - (double)k);
}

/* quadratic_form :133-142 */
static double oc_quadratic_form(const double *co, int m, int k, double sigma, const double *x) {
    double norm_squared = 0.0;
    for (int r = 0; r < m; ++r) norm_squared += x[r] * x[r];
    double *t = (double *)malloc(sizeof(double) * (k + 1));
    double *inv = (double *)malloc(sizeof(double) * (k * k + 1));
    double *ti = (double *)malloc(sizeof(double) * (k + 1));
    for (int a = 0; a < k; ++a) {
        double s = 0.0;
        for (int r = 0; r < m; ++r) s += co[r * k + a] * x[r];
        t[a] = s;
    }
    oc_inner_inverse(co, m, k, sigma, inv);
    for (int b = 0; b < k; ++b) { /* (t^T * inv) */
        double s = 0.0;
        for (int a = 0; a < k; ++a) s += t[a] * inv[a * k + b];
        ti[b] = s;
    }
    double q = 0.0;
    for (int b = 0; b < k; ++b) q += ti[b] * t[b];
    free(t); free(inv); free(ti);
    return (norm_squared - q) / (sigma * sigma);
}

/* Exposed for the reference's two KATs (ppca_model.rs:658-671): full (unmasked) C. */
double ppca_oracle_quadratic_form(const double *c, int d, int k, double sigma, const double *x) {
    return oc_quadratic_form(c, d, k, sigma, x);
}
double ppca_oracle_covariance_log_det(const double *c, int d, int k, double sigma) {
    return oc_covariance_log_det(c, d, k, sigma);
}

/* ---------------------------------------------------------- sample helpers */

/* OutputCovariance::masked :123-131 + Mask::mask utils.rs:56-61.
 * Gathers observed rows of C into co (m x k) and (x - mean) into sub (m).
 * A position is observed iff x is finite (dataset.rs:19-22).  Returns m. */
static int gather_observed(const double *x, const double *c, const double *mean, int d, int k,
                           double *co, double *sub) {
    int m = 0;
    for (int j = 0; j < d; ++j) {
        if (is_finite(x[j])) {
            if (co) memcpy(co + (size_t)m * k, c + (size_t)j * k, sizeof(double) * k);
            if (sub) sub[m] = x[j] - mean[j];
            ++m;
        }
    }
    return m;
}

/* llk_one ppca_model.rs:124-139 */
static double llk_one(const double *x, int d, int k, double sigma, const double *c, const double *mean,
                      double *co, double *sub) {
    int m = gather_observed(x, c, mean, d, k, co, sub);
    if (m == 0) return 0.0; /* :125-129 */
    return -oc_quadratic_form(co, m, k, sigma, sub) / 2.0
           - oc_covariance_log_det(co, m, k, sigma) / 2.0
           - LN_2PI / 2.0 * (double)m;
}

/* llks ppca_model.rs:152-159 (per sample, unweighted) */
void ppca_oracle_llks(const double *x, int64_t n, int d, int k, double sigma, const double *c,
                      const double *mean, double *out) {
#pragma omp parallel
    {
        double *co = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
        double *sub = (double *)malloc(sizeof(double) * (d + 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) out[i] = llk_one(x + i * d, d, k, sigma, c, mean, co, sub);
        free(co); free(sub);
    }
}

/* llk ppca_model.rs:142-149 (weighted sum; fixed summation order here) */
double ppca_oracle_llk(const double *x, const double *w, int64_t n, int d, int k, double sigma,
                       const double *c, const double *mean) {
    double *l = (double *)malloc(sizeof(double) * (n + 1));
    ppca_oracle_llks(x, n, d, k, sigma, c, mean, l);
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += l[i] * (w ? w[i] : 1.0);
    free(l);
    return s;
}

/* infer_one ppca_model.rs:195-208 (uninferred :98-104 for all-masked samples) */
static void infer_one(const double *x, int d, int k, double sigma, const double *c, const double *mean,
                      double *co, double *sub, double *e, double *state, double *cov) {
    int m = gather_observed(x, c, mean, d, k, co, sub);
    if (m == 0) {
        for (int a = 0; a < k; ++a) state[a] = 0.0;
        for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) cov[a * k + b] = (a == b) ? 1.0 : 0.0;
        return;
    }
    oc_estimator_transform(co, m, k, sigma, e);
    for (int a = 0; a < k; ++a) {
        double s = 0.0;
        for (int r = 0; r < m; ++r) s += e[a * m + r] * sub[r];
        state[a] = s;
    }
    oc_estimator_covariance(co, m, k, sigma, cov);
}

/* infer ppca_model.rs:221-227: states (n x k), covs (n x k x k) */
void ppca_oracle_infer(const double *x, int64_t n, int d, int k, double sigma, const double *c,
                       const double *mean, double *states, double *covs) {
#pragma omp parallel
    {
        double *co = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
        double *sub = (double *)malloc(sizeof(double) * (d + 1));
        double *e = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
        double *cv = (double *)malloc(sizeof(double) * (k * k + 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            infer_one(x + i * d, d, k, sigma, c, mean, co, sub, e, states + i * k, covs ? covs + i * k * k : cv);
        }
        free(co); free(sub); free(e); free(cv);
    }
}

/* smooth :237-244 (mode 0) / extrapolate :254-261 (mode 1);
 * smoothed :454-456, extrapolated :460-463, Mask::choose utils.rs:137-153 */
void ppca_oracle_reconstruct(const double *x, int64_t n, int d, int k, double sigma, const double *c,
                             const double *mean, int mode, double *out) {
    double *states = (double *)malloc(sizeof(double) * ((size_t)n * k + 1));
    ppca_oracle_infer(x, n, d, k, sigma, c, mean, states, NULL);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            for (int a = 0; a < k; ++a) s += c[(size_t)j * k + a] * states[i * k + a];
            s += mean[j];
            out[i * d + j] = (mode == 1 && is_finite(x[i * d + j])) ? x[i * d + j] : s;
        }
    }
    free(states);
}

/* ------------------------------------------------------------------ prior */

typedef struct {
    int has_mean_prior;            /* prior.rs:93-95 */
    const double *mean;            /* d */
    const double *mean_covariance; /* d x d; precision = try_inverse (prior.rs:36-41) */
    int has_isotropic_noise_prior; /* prior.rs:71-73 */
    double isotropic_noise_alpha, isotropic_noise_beta;
    double transformation_precision; /* prior.rs:89-91 */
} ppca_oracle_prior;

/* Prior::smooth_mean prior.rs:97-110.  precision = diag(pdiag).  0 on failure. */
static int prior_smooth_mean(const ppca_oracle_prior *p, int d, double *mean, const double *pdiag) {
    double *pp = (double *)malloc(sizeof(double) * ((size_t)d * d + 1));
    double *tot = (double *)malloc(sizeof(double) * ((size_t)d * d + 1));
    double *num = (double *)malloc(sizeof(double) * (d + 1));
    double *x = (double *)malloc(sizeof(double) * (d + 1));
    int ok = mat_inverse(p->mean_covariance, d, pp);
    if (ok) {
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) tot[i * d + j] = pp[i * d + j] + (i == j ? pdiag[i] : 0.0);
        for (int i = 0; i < d; ++i) {
            double s = 0.0;
            for (int j = 0; j < d; ++j) s += pp[i * d + j] * p->mean[j];
            num[i] = s + pdiag[i] * mean[i];
        }
        ok = qr_solve(tot, d, num, x);
        if (ok) memcpy(mean, x, sizeof(double) * d);
    }
    free(pp); free(tot); free(num); free(x);
    return ok;
}

/* ------------------------------------------------------ iterate_with_prior */

/* ppca_model.rs:277-393.  w may be NULL (all 1.0).  prior may be NULL (default).
 * Outputs: c_out (d x k), mean_out (d), *sigma_out.  Returns 0 on success,
 * -1 if no sample has an observed value (the reference panics at :358),
 * -2 if the mean-prior system cannot be solved. */
int ppca_oracle_iterate(const double *x, const double *w, int64_t n, int d, int k, double sigma,
                        const double *c, const double *mean, const ppca_oracle_prior *prior,
                        double *c_out, double *mean_out, double *sigma_out) {
    const int kk = k * k;
    /* :278  let inferred = self.infer(dataset);  -- materialised, as upstream */
    double *states = (double *)malloc(sizeof(double) * ((size_t)n * k + 1));
    double *covs = (double *)malloc(sizeof(double) * ((size_t)n * kk + 1));
    ppca_oracle_infer(x, n, d, k, sigma, c, mean, states, covs);

    int nthreads = ppca_oracle_num_threads();

    /* :281-293  total_cross_moment = sum_i w_i * fillna(x_i - mean) * state_i^T  (d x k) */
    double *cross = (double *)calloc((size_t)d * k + 1, sizeof(double));
    {
        double *part = (double *)calloc((size_t)nthreads * d * k + 1, sizeof(double));
#pragma omp parallel
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            double *acc = part + (size_t)t * d * k;
#pragma omp for schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                double wi = w ? w[i] : 1.0;
                const double *xi = x + i * d;
                for (int j = 0; j < d; ++j) {
                    double cf = is_finite(xi[j]) ? xi[j] - mean[j] : 0.0; /* fillna utils.rs:118-127 */
                    double wc = wi * cf;
                    for (int a = 0; a < k; ++a) acc[(size_t)j * k + a] += wc * states[i * k + a];
                }
            }
        }
        for (int t = 0; t < nthreads; ++t)
            for (size_t e = 0; e < (size_t)d * k; ++e) cross[e] += part[(size_t)t * d * k + e];
        free(part);
    }

    /* :294-325  per-dimension sequential scans, parallel over d */
    double tp = prior ? prior->transformation_precision : 0.0;
#pragma omp parallel
    {
        double *sm = (double *)malloc(sizeof(double) * (kk + 1));
        double *sol = (double *)malloc(sizeof(double) * (k + 1));
#pragma omp for schedule(dynamic, 1)
        for (int idx = 0; idx < d; ++idx) {
            for (int e = 0; e < kk; ++e) sm[e] = 0.0;
            for (int64_t i = 0; i < n; ++i) {
                if (!is_finite(x[i * d + idx])) continue; /* :302 */
                double wi = w ? w[i] : 1.0;
                const double *s = states + i * k;
                const double *cv = covs + i * kk;
                /* :303  weight * inferred.second_moment()  (:437-439) */
                for (int a = 0; a < k; ++a)
                    for (int b = 0; b < k; ++b) sm[a * k + b] += wi * (s[a] * s[b] + cv[a * k + b]);
            }
            for (int a = 0; a < k; ++a) sm[a * k + a] += tp * 1.0; /* :307-308 */
            if (qr_solve(sm, k, cross + (size_t)idx * k, sol)) {
                memcpy(c_out + (size_t)idx * k, sol, sizeof(double) * k);
            } else {
                memcpy(c_out + (size_t)idx * k, c + (size_t)idx * k, sizeof(double) * k); /* :313-321 */
            }
        }
        free(sm); free(sol);
    }

    /* :328-358  (square_error, deviations_square_sum, total_deviation, totals) */
    double square_error = 0.0, deviations_square_sum = 0.0;
    double *total_deviation = (double *)calloc(d + 1, sizeof(double));
    double *totals = (double *)calloc(d + 1, sizeof(double));
    int64_t n_nonempty = 0;
    {
        double *pdev = (double *)calloc((size_t)nthreads * d + 1, sizeof(double));
        double *ptot = (double *)calloc((size_t)nthreads * d + 1, sizeof(double));
        double *psq = (double *)calloc(nthreads + 1, sizeof(double));
        double *pds = (double *)calloc(nthreads + 1, sizeof(double));
        int64_t *pne = (int64_t *)calloc(nthreads + 1, sizeof(int64_t));
#pragma omp parallel
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            double *co = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
            double *cs = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
#pragma omp for schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                const double *xi = x + i * d;
                int m = gather_observed(xi, c, mean, d, k, co, NULL); /* :336 */
                if (m == 0) continue;                                 /* :333 */
                double wi = w ? w[i] : 1.0;
                const double *s = states + i * k;
                const double *cv = covs + i * kk;
                /* :345  (sub_transform * covariance).dot(sub_transform) */
                double dot = 0.0;
                for (int r = 0; r < m; ++r)
                    for (int b = 0; b < k; ++b) {
                        double v = 0.0;
                        for (int a = 0; a < k; ++a) v += co[(size_t)r * k + a] * cv[a * k + b];
                        cs[(size_t)r * k + b] = v;
                    }
                for (int r = 0; r < m; ++r)
                    for (int b = 0; b < k; ++b) dot += cs[(size_t)r * k + b] * co[(size_t)r * k + b];
                /* :338-342  deviation = fillna(x - C*state - mean) */
                double ns = 0.0;
                for (int j = 0; j < d; ++j) {
                    if (!is_finite(xi[j])) continue;
                    double cz = 0.0;
                    for (int a = 0; a < k; ++a) cz += c[(size_t)j * k + a] * s[a];
                    double dev = xi[j] - cz - mean[j];
                    ns += dev * dev;
                    pdev[(size_t)t * d + j] += wi * dev;
                    ptot[(size_t)t * d + j] += wi * 1.0; /* as_vector utils.rs:129-135 */
                }
                psq[t] += wi * dot;
                pds[t] += wi * ns;
                pne[t] += 1;
            }
            free(co); free(cs);
        }
        for (int t = 0; t < nthreads; ++t) {
            square_error += psq[t];
            deviations_square_sum += pds[t];
            n_nonempty += pne[t];
            for (int j = 0; j < d; ++j) { total_deviation[j] += pdev[(size_t)t * d + j]; totals[j] += ptot[(size_t)t * d + j]; }
        }
        free(pdev); free(ptot); free(psq); free(pds); free(pne);
    }
    int rc = 0;
    if (n_nonempty == 0) rc = -1; /* :358 expect("non-empty dataset") */

    double totals_sum = 0.0;
    for (int j = 0; j < d; ++j) totals_sum += totals[j];
    double isotropic_noise_sq;
    if (prior && prior->has_isotropic_noise_prior) { /* :360-368 */
        isotropic_noise_sq = ((square_error + deviations_square_sum) / 2.0 + prior->isotropic_noise_beta)
                             / (totals_sum / 2.0 + prior->isotropic_noise_alpha + 1.0);
    } else {
        isotropic_noise_sq = (square_error + deviations_square_sum) / totals_sum; /* :370 */
    }
    /* :373-377 */
    for (int j = 0; j < d; ++j)
        mean_out[j] = (totals[j] > 0.0 ? total_deviation[j] / totals[j] : 0.0) + mean[j];
    if (rc == 0 && prior && prior->has_mean_prior) { /* :379-384 */
        double *pdiag = (double *)malloc(sizeof(double) * (d + 1));
        for (int j = 0; j < d; ++j) pdiag[j] = totals[j] / isotropic_noise_sq;
        if (!prior_smooth_mean(prior, d, mean_out, pdiag)) rc = -2;
        free(pdiag);
    }
    *sigma_out = sqrt(isotropic_noise_sq); /* :389 */

    free(states); free(covs); free(cross); free(total_deviation); free(totals);
    return rc;
}

/* The packed sufficient statistics of one shard, in the layout of
 * include/ppca_hip.h (ppca_stats_len).  Computed here in the reference's own
 * per-sample quantities (states/covariances from infer_one), so that the
 * product's host finalisation and the multi-rank reduce can be checked on CPU.
 *   cross[d*k] | S[d*k'] (lower-packed) | U[d*k] | sumx[d] | totals[d] |
 *   sqerr, devsq, llk, sumw, n_nonempty, 0,0,0
 */
void ppca_oracle_stats(const double *x, const double *w, int64_t n, int d, int k, double sigma,
                       const double *c, const double *mean, double *stats) {
    const int kk = k * k, kp = k * (k + 1) / 2;
    size_t o_cross = 0, o_s = (size_t)d * k, o_u = o_s + (size_t)d * kp, o_sx = o_u + (size_t)d * k,
           o_tot = o_sx + d, o_sc = o_tot + d, len = o_sc + 8;
    for (size_t e = 0; e < len; ++e) stats[e] = 0.0;
    double *states = (double *)malloc(sizeof(double) * ((size_t)n * k + 1));
    double *covs = (double *)malloc(sizeof(double) * ((size_t)n * kk + 1));
    double *llks = (double *)malloc(sizeof(double) * (n + 1));
    double *co = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
    ppca_oracle_infer(x, n, d, k, sigma, c, mean, states, covs);
    ppca_oracle_llks(x, n, d, k, sigma, c, mean, llks);
    for (int64_t i = 0; i < n; ++i) {
        const double *xi = x + i * d;
        double wi = w ? w[i] : 1.0;
        const double *s = states + i * k;
        const double *cv = covs + i * kk;
        int m = 0;
        double ns = 0.0;
        for (int j = 0; j < d; ++j) {
            if (!is_finite(xi[j])) continue;
            memcpy(co + (size_t)m * k, c + (size_t)j * k, sizeof(double) * k);
            ++m;
            double cf = xi[j] - mean[j];
            double cz = 0.0;
            for (int a = 0; a < k; ++a) cz += c[(size_t)j * k + a] * s[a];
            ns += (cf - cz) * (cf - cz);
            for (int a = 0; a < k; ++a) stats[o_cross + (size_t)j * k + a] += wi * cf * s[a];
            int e = 0;
            for (int a = 0; a < k; ++a)
                for (int b = 0; b <= a; ++b, ++e)
                    stats[o_s + (size_t)j * kp + e] += wi * (s[a] * s[b] + 0.5 * (cv[a * k + b] + cv[b * k + a]));
            for (int a = 0; a < k; ++a) stats[o_u + (size_t)j * k + a] += wi * s[a];
            stats[o_sx + j] += wi * cf;
            stats[o_tot + j] += wi;
        }
        stats[o_sc + 2] += wi * llks[i];
        stats[o_sc + 3] += wi;
        if (m == 0) continue;
        double dot = 0.0;
        for (int r = 0; r < m; ++r)
            for (int b = 0; b < k; ++b) {
                double v = 0.0;
                for (int a = 0; a < k; ++a) v += co[(size_t)r * k + a] * cv[a * k + b];
                dot += v * co[(size_t)r * k + b];
            }
        stats[o_sc + 0] += wi * dot;
        stats[o_sc + 1] += wi * ns;
        stats[o_sc + 4] += 1.0;
    }
    free(states); free(covs); free(llks); free(co);
}

/* An honestly optimised CPU form of the same statistics pass (SURVEY.md 8d asks for it beside the literal
 * port so that the GPU/CPU ratio is not only against the reference's structure): ONE sweep over the samples,
 * per-sample Gram from a precomputed vech(c_j c_j^T) table, Cholesky instead of the subtractive Woodbury form,
 * packed second moments, thread-private statistics summed at the end.  Same output layout as
 * ppca_oracle_stats.  Not a restatement of any reference function: a baseline, checked against
 * ppca_oracle_stats in tests/test_oracle.py. */
void ppca_oracle_fused_stats(const double *x, const double *w, int64_t n, int d, int k, double sigma,
                             const double *c, const double *mean, double *stats) {
    const int kp = k * (k + 1) / 2;
    const size_t o_cross = 0, o_s = (size_t)d * k, o_u = o_s + (size_t)d * kp, o_sx = o_u + (size_t)d * k,
                 o_tot = o_sx + d, o_sc = o_tot + d, len = o_sc + 8;
    const double s2 = sigma * sigma, ln_sigma = log(sigma), ln_2pi = 1.8378770664093453;
    double *q = (double *)malloc(sizeof(double) * ((size_t)d * kp + 1));
    for (int j = 0; j < d; ++j) {
        int e = 0;
        for (int a = 0; a < k; ++a)
            for (int b = 0; b <= a; ++b, ++e) q[(size_t)j * kp + e] = c[(size_t)j * k + a] * c[(size_t)j * k + b];
    }
    for (size_t e = 0; e < len; ++e) stats[e] = 0.0;
#pragma omp parallel
    {
        double *loc = (double *)calloc(len, sizeof(double));
        double *g = (double *)malloc(sizeof(double) * (size_t)(2 * kp + 3 * k + 1));
        double *l = g + kp, *b = l + kp, *z = b + k, *u = z + k;
        int *obs = (int *)malloc(sizeof(int) * (size_t)(d + 1));
        double *xt = (double *)malloc(sizeof(double) * (size_t)(d + 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const double *xi = x + i * d;
            const double wi = w ? w[i] : 1.0;
            int m = 0;
            double xx = 0.0;
            for (int j = 0; j < d; ++j)
                if (is_finite(xi[j])) {
                    obs[m] = j;
                    xt[m] = xi[j] - mean[j];
                    xx += xt[m] * xt[m];
                    ++m;
                }
            loc[o_sc + 3] += wi;
            for (int e = 0; e < kp; ++e) g[e] = 0.0;
            for (int a = 0; a < k; ++a) b[a] = 0.0;
            for (int r = 0; r < m; ++r) {
                const double *qj = q + (size_t)obs[r] * kp, *cj = c + (size_t)obs[r] * k;
                for (int e = 0; e < kp; ++e) g[e] += qj[e];
                for (int a = 0; a < k; ++a) b[a] += cj[a] * xt[r];
            }
            /* M = G + s2 I = L L^T (packed lower), ln det M */
            double logdet = 0.0;
            for (int a = 0; a < k; ++a)
                for (int cc = 0; cc <= a; ++cc) {
                    double sacc = g[a * (a + 1) / 2 + cc] + (a == cc ? s2 : 0.0);
                    for (int t = 0; t < cc; ++t) sacc -= l[a * (a + 1) / 2 + t] * l[cc * (cc + 1) / 2 + t];
                    if (a == cc) {
                        logdet += log(sacc);
                        l[a * (a + 1) / 2 + a] = 1.0 / sqrt(sacc);
                    } else {
                        l[a * (a + 1) / 2 + cc] = sacc * l[cc * (cc + 1) / 2 + cc];
                    }
                }
            double quad = 0.0, zz = 0.0;
            for (int a = 0; a < k; ++a) {
                double sacc = b[a];
                for (int t = 0; t < a; ++t) sacc -= l[a * (a + 1) / 2 + t] * z[t];
                z[a] = sacc * l[a * (a + 1) / 2 + a];
                quad += z[a] * z[a];
            }
            for (int a = k - 1; a >= 0; --a) {
                double sacc = z[a];
                for (int t = a + 1; t < k; ++t) sacc -= l[t * (t + 1) / 2 + a] * z[t];
                z[a] = sacc * l[a * (a + 1) / 2 + a];
                zz += z[a] * z[a];
            }
            /* g <- w (z z^T + s2 M^-1), packed; trace of M^-1 on the way */
            double tr = 0.0;
            for (int cc = 0; cc < k; ++cc) {
                for (int a = cc; a < k; ++a) {
                    double sacc = (a == cc) ? 1.0 : 0.0;
                    for (int t = cc; t < a; ++t) sacc -= l[a * (a + 1) / 2 + t] * u[t];
                    u[a] = sacc * l[a * (a + 1) / 2 + a];
                }
                for (int a = k - 1; a >= cc; --a) {
                    double sacc = u[a];
                    for (int t = a + 1; t < k; ++t) sacc -= l[t * (t + 1) / 2 + a] * u[t];
                    u[a] = sacc * l[a * (a + 1) / 2 + a];
                    g[a * (a + 1) / 2 + cc] = wi * (z[a] * z[cc] + s2 * u[a]);
                }
                tr += u[cc];
            }
            for (int a = 0; a < k; ++a) b[a] = wi * z[a];
            for (int r = 0; r < m; ++r) {
                const int j = obs[r];
                double *sj = loc + o_s + (size_t)j * kp, *uj = loc + o_u + (size_t)j * k, *cj = loc + o_cross + (size_t)j * k;
                for (int e = 0; e < kp; ++e) sj[e] += g[e];
                for (int a = 0; a < k; ++a) {
                    uj[a] += b[a];
                    cj[a] += b[a] * xt[r];
                }
                loc[o_sx + j] += wi * xt[r];
                loc[o_tot + j] += wi;
            }
            if (m > 0) {
                loc[o_sc + 0] += wi * s2 * ((double)k - s2 * tr);           /* tr(C_o Sigma C_o^T) */
                loc[o_sc + 1] += wi * (xx - quad - s2 * zz);                 /* |x~ - C_o z|^2 */
                loc[o_sc + 2] += wi * -0.5 * ((xx - quad) / s2 + logdet + 2.0 * ln_sigma * (double)(m - k) + ln_2pi * (double)m);
                loc[o_sc + 4] += 1.0;
            }
        }
#pragma omp critical
        for (size_t e = 0; e < len; ++e) stats[e] += loc[e];
        free(loc); free(g); free(obs); free(xt);
    }
    free(q);
}

/* ------------------------------------------------------------ to_canonical */

/* ppca_model.rs:398-425: C = U S V^T -> C' = U S, columns by descending singular
 * value, each multiplied by signum(sum(column)).  One-sided Jacobi: rotating the
 * columns of C until they are orthogonal yields C*V = U*S directly. */
void ppca_oracle_to_canonical(const double *c, int d, int k, double *out) {
    if (k == 0) return; /* :400-402 */
    double *a = (double *)malloc(sizeof(double) * ((size_t)d * k + 1));
    memcpy(a, c, sizeof(double) * (size_t)d * k);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int r = 0; r < d; ++r) {
                    app += a[(size_t)r * k + p] * a[(size_t)r * k + p];
                    aqq += a[(size_t)r * k + q] * a[(size_t)r * k + q];
                    apq += a[(size_t)r * k + p] * a[(size_t)r * k + q];
                }
                if (apq == 0.0) continue;
                double scale = sqrt(app * aqq);
                if (scale > 0.0 && fabs(apq) / scale > off) off = fabs(apq) / scale;
                double zeta = (aqq - app) / (2.0 * apq);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int r = 0; r < d; ++r) {
                    double vp = a[(size_t)r * k + p], vq = a[(size_t)r * k + q];
                    a[(size_t)r * k + p] = cs * vp - sn * vq;
                    a[(size_t)r * k + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
    /* order columns by descending norm (= singular value) */
    double *nrm = (double *)malloc(sizeof(double) * (k + 1));
    int *ord = (int *)malloc(sizeof(int) * (k + 1));
    for (int p = 0; p < k; ++p) {
        double s = 0;
        for (int r = 0; r < d; ++r) s += a[(size_t)r * k + p] * a[(size_t)r * k + p];
        nrm[p] = s; ord[p] = p;
    }
    for (int p = 0; p < k; ++p)
        for (int q = p + 1; q < k; ++q)
            if (nrm[ord[q]] > nrm[ord[p]]) { int t = ord[p]; ord[p] = ord[q]; ord[q] = t; }
    for (int p = 0; p < k; ++p) {
        int src = ord[p];
        double sum = 0;
        for (int r = 0; r < d; ++r) sum += a[(size_t)r * k + src];
        double sg = isnan(sum) ? sum : (signbit(sum) ? -1.0 : 1.0); /* f64::signum: +0.0 -> 1.0 */
        for (int r = 0; r < d; ++r) out[(size_t)r * k + p] = a[(size_t)r * k + src] * sg;
    }
    free(a); free(nrm); free(ord);
}

/* --------------------------------------------------------------- mixture */

/* robust_log_softmax mix.rs:14-18 (in place on v[0..n)) */
static void robust_log_softmax(double *v, int n) {
    double mx = v[0];
    for (int i = 1; i < n; ++i) if (v[i] > mx) mx = v[i];
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(v[i] - mx);
    double ln = log(s);
    for (int i = 0; i < n; ++i) v[i] = v[i] - mx - ln;
}

/* robust_log_softnorm mix.rs:21-25 */
static double robust_log_softnorm(const double *v, int n) {
    double mx = v[0];
    for (int i = 1; i < n; ++i) if (v[i] > mx) mx = v[i];
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(v[i] - mx);
    return mx + log(s);
}

/* Mixture model = n_models PPCA models with a common d and k (the reference
 * allows per-model k; the parity configs use one k).  Parameters are packed:
 * sigmas[M], cs[M][d*k], means[M][d], log_weights[M]. */

/* PPCAMix::llks mix.rs:152-159 (per sample) via llk_one :147-149 */
void ppca_oracle_mix_llks(const double *x, int64_t n, int d, int k, int nm, const double *sigmas,
                          const double *cs, const double *means, const double *log_weights, double *out) {
    double *l = (double *)malloc(sizeof(double) * ((size_t)nm * n + 1));
    for (int c = 0; c < nm; ++c)
        ppca_oracle_llks(x, n, d, k, sigmas[c], cs + (size_t)c * d * k, means + (size_t)c * d, l + (size_t)c * n);
    double *v = (double *)malloc(sizeof(double) * ((size_t)nm + 1));
    for (int64_t i = 0; i < n; ++i) {
        for (int c = 0; c < nm; ++c) v[c] = l[(size_t)c * n + i] + log_weights[c];
        out[i] = robust_log_softnorm(v, nm);
    }
    free(v);
    free(l);
}

/* PPCAMix::infer_cluster mix.rs:179-189: log posteriors (n x nm) */
void ppca_oracle_mix_infer_cluster(const double *x, int64_t n, int d, int k, int nm, const double *sigmas,
                                   const double *cs, const double *means, const double *log_weights,
                                   double *out) {
    double *l = (double *)malloc(sizeof(double) * ((size_t)nm * n + 1));
    for (int c = 0; c < nm; ++c)
        ppca_oracle_llks(x, n, d, k, sigmas[c], cs + (size_t)c * d * k, means + (size_t)c * d, l + (size_t)c * n);
    for (int64_t i = 0; i < n; ++i) {
        double *v = out + i * nm;
        for (int c = 0; c < nm; ++c) v[c] = l[(size_t)c * n + i] + log_weights[c];
        robust_log_softmax(v, nm);
    }
    free(l);
}

/* PPCAMix::iterate_with_prior mix.rs:281-337.
 * DIVERGENCE (documented in DESIGN.md): the reference drops samples with
 * w_i <= 0 from the weight vector only (:304-309) and then pairs the shortened
 * vector with the unfiltered data (:326), which misaligns weights.  The oracle
 * requires all w_i > 0 (returns -3 otherwise), where both agree. */
int ppca_oracle_mix_iterate(const double *x, const double *w, int64_t n, int d, int k, int nm,
                            const double *sigmas, const double *cs, const double *means,
                            const double *log_weights, const ppca_oracle_prior *prior,
                            double *sigmas_out, double *cs_out, double *means_out, double *log_weights_out) {
    for (int64_t i = 0; i < n; ++i) if (w && !(w[i] > 0.0)) return -3;
    double *lp = (double *)malloc(sizeof(double) * ((size_t)n * nm + 1));
    ppca_oracle_mix_infer_cluster(x, n, d, k, nm, sigmas, cs, means, log_weights, lp); /* :283-295 */
    double *u = (double *)malloc(sizeof(double) * (n + 1));
    int rc = 0;
    for (int c = 0; c < nm && rc == 0; ++c) {
        double mx = -INFINITY;
        for (int64_t i = 0; i < n; ++i) { /* :304-317 */
            double v = log(w ? w[i] : 1.0) + lp[i * nm + c];
            u[i] = v;
            if (!isnan(v) && v > mx) mx = v;
        }
        double sum = 0.0;
        for (int64_t i = 0; i < n; ++i) { u[i] = exp(u[i] - mx); sum += u[i]; } /* :320-325 */
        log_weights_out[c] = log(sum) + mx;
        rc = ppca_oracle_iterate(x, u, n, d, k, sigmas[c], cs + (size_t)c * d * k, means + (size_t)c * d, prior,
                                 cs_out + (size_t)c * d * k, means_out + (size_t)c * d, sigmas_out + c); /* :326-328 */
    }
    robust_log_softmax(log_weights_out, nm); /* :335 */
    free(lp); free(u);
    return rc;
}
