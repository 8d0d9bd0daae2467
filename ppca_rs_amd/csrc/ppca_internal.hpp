// ppca_internal.hpp -- declarations shared by the kernels (ppca_kernels.hip) and
// the C-ABI host layer (ppca_capi.hip).  Not installed; the public surface is
// include/ppca_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "ppca_small.hpp"

namespace ppca {

// Device model buffer: [sigma, sigma^2, ln sigma, 0 | C (d x k row-major) | mean (d)]
constexpr int MODEL_HDR = 4;
inline int64_t model_len(int d, int k) { return MODEL_HDR + (int64_t)d * k + d; }

constexpr int FUSED_MAX_D = 256;  // one LDS-resident tile of 32 samples x 256 dims
constexpr int FUSED_MAX_K = 10;   // k(k+1)/2 + k + 1 <= 80 accumulator columns
constexpr int FUSED_TILE = 32;    // samples per tile
constexpr int FUSED_THREADS = 256;

struct PassArgs {
    const double *X;      // n x d row-major, non-finite = masked
    int64_t ldx;          // row stride in elements
    const double *w;      // n weights or nullptr (= 1)
    const int *rows;      // EM mode, nullable: gathered pass -- sample i is physical row rows[i] of X (w is indexed by i)
    int64_t n;
    const int *n_dev;     // nullable: the number of samples is read from device memory instead of n (gathered passes whose
                          // row count was produced on the device: no host round trip); the grid is sized for an upper bound
    int d;
    const double *model;  // device model buffer
    // EM mode
    double *part;         // [grid][stats_len] per-workgroup partial statistics
    // post mode (all nullable)
    double *scal_part;    // [grid][8] per-workgroup scalars (llk, sumw, ...)
    double *llks;         // n
    double *states;       // n x k
    double *covs;         // n x k x k
    double *recon;        // n x d
    int recon_mode;       // 0 smooth, 1 extrapolate, 2 smoothed cov diag, 3 extrapolated cov diag
    signed char *qtab;    // int8 Gram slice table of the current model (written by the launcher's qprep)
    double *qscale;       // its 64 dequantisation multipliers
    int *qflag;           // per column tile: 1 = the int8 form is not safe for this model (qprep's dynamic-range
                          // guard) -> the fp64-Gram instantiation runs; nullptr = run unconditionally
    int no_llk;           // EM mode: the caller does not read SC_LLK (mixture component steps): skip the
                          // per-sample logarithm of the weighted path
    double *dbg;          // diagnostic builds (-DPPCA_PHASE_TIMING): [grid][4] phase cycle sums
    // EM mode, the guard of the int8 form of the mask-side statistics (launch_em_wguard):
    double *errb;         // [grid][W_GUARD_NCOL]: per workgroup and column of [wP | wz | w], a bound of the rounding the fixed-point
                          // cut added to any sum of that column (written by em8_kernel; nullptr: not collected)
    const double *cpad;   // zero-padded copy of C, [256][k + 1], written by qprep_kernel (em8's -DE8_C_GLOBAL experiment)
    const double *cpb;    // C as the A operands of em9_kernel's b = X~ C on v_mfma_f64_4x4x4, in operand order (CPB_DOUBLES, written
                          // by qprep_kernel: see cpb_index)
    const int *runflag;   // nullable: second stage of a guarded EM pass (launch_em_fallback): the fp64 instantiation of pass_kernel
                          // reads the verdict of reduce_wguard_kernel: 0 = return at once; 1 = the whole pass again; 2 = only the
                          // slices (runs of tiles, as the guarded launch dealt them) of the flagged workgroups `who`, each slice
                          // split over gridDim.x / n_flagged workgroups; nullptr: qflag decides as before
    const int *who;       // [n_flagged] (runflag[QF_NFLAGGED - QF_MODE]) ascending workgroup indices of the guarded launch
    int heavy_max;        // em9_kernel's back role: a tile with at most this many rows that do not fit the fixed-point form of the
                          // mask-side statistics sends those rows round it (exact fp64 adds) instead of raising its exponents; 0: every
                          // such tile raises them (rounds 3-5: ppca_ctx_set_heavy_rows, a test hook)
    int skip_qprep;       // the slice table / guard flags / padded C behind qtab are already those of `model` (written by
                          // finalize_qprep_kernel at the end of the previous EM step): the pass launches no qprep_kernel
};
// Guard words behind PassArgs::qflag (ints): [0, 8) the Gram guard's per-tile flags (qprep_kernel); [8] the fallback's mode (see
// PassArgs::runflag); [9] the verdict of the W-side check; [10] ticket counter of reduce_wguard_kernel's last-block election;
// [11] workgroups whose slices the fallback recomputes; [12] rows in those slices; [13] the Gram guard's verdict for the pass (the tile
// flags themselves are overwritten by the NEXT model's when the step's finalisation builds its table).
constexpr int QF_MODE = 8, QF_WVERDICT = 9, QF_TICKET = 10, QF_NFLAGGED = 11, QF_NROWS2 = 12, QF_GVERDICT = 13;
// What reduce_wguard_kernel needs beside the partials.
struct GuardArgs {
    const double *errb;   // [grid][W_GUARD_NCOL] rounding bounds of the workgroups' cuts; nullptr: no W-side check (kernels that
                          // do not cut their rows: only the Gram flags decide)
    double *es;           // [W_GUARD_NCOL] scratch: their column sums (written by the reduction's extra workgroups)
    int *qflag;           // the guard words above
    int *wgflag;          // [grid]: 1 = the workgroup's partial is replaced by the fallback's
    int *who;             // [grid]: the flagged workgroups, ascending (PassArgs::who of the fallback)
    int64_t n;            // the guarded pass's row count (n_dev nullable: read from the device)
    const int *n_dev;
    int d;
    int grid;             // workgroups of the guarded launch (= partials)
};
// Multi-component launches of the mixture step (round 6): the per-component pointers travel BY VALUE in the kernel arguments, so a
// launch covers at most MIX_MAX components (larger mixtures, or components of different state sizes, run component by component).
constexpr int MIX_MAX = 16;
struct MixLlkArgs {  // mix_llk8_kernel (ppca_llk.hip): the llk sweeps of all components in one launch
    const double *X;
    int64_t ldx, n;
    int d, nm;
    int runs_per_xcd;             // mix_llk_runs_per_xcd(grid, nm)
    const double *model[MIX_MAX];
    void *tab[MIX_MAX];           // the component's slice-table block (fused_qtab_view)
    double *llks[MIX_MAX];        // n per component
};
struct MixTabArgs {  // qprep_multi_kernel: slice tables + guard flags + padded C of nm models in one launch
    int d, nm;
    const double *model[MIX_MAX];
    void *tab[MIX_MAX];
};
struct MixFinalArgs {  // finalize_qprep_multi_kernel: the M-step finalisation of nm components + the new models' tables
    int d, nm;
    double tau;
    int has_ig;
    double alpha, beta;
    const double *stats[MIX_MAX];
    const double *min[MIX_MAX];
    double *mout[MIX_MAX];
    void *tab[MIX_MAX];
};
struct MixReduceArgs {  // reduce_wguard_multi_kernel: reduction + verdict of nm guarded EM passes in one launch
    int nm;
    int64_t len;
    const double *part[MIX_MAX];
    double *out[MIX_MAX];
    GuardArgs g[MIX_MAX];
};

// em9_kernel's b = X~ C on v_mfma_f64_4x4x4 (round 5): one instruction = four blocks = (two
// 4-dim groups kb) x (two 4-sample groups sb); its A operand is C^T: lane 16 k + 8 kb + 4 sb + i holds C[dim][4 c + i] with
// dim = 128 kq + 32 (q >> 2) + 16 kb + 4 (q & 3) + k for the wave's dimension half kq, step q = 0..15 and column group c -- the same
// value for sb = 0, 1, so a (kq, q, c) operand is 32 doubles, stored contiguously: entry (k, kb, i) at (2 k + kb) 4 + i.
constexpr int CPB_GROUPS = 3;  // column groups of 4 at k = FUSED_MAX_K = 10
constexpr int CPB_DOUBLES = 2 * 16 * CPB_GROUPS * 32;
constexpr int W_GUARD_NCOL = 80;  // 16 x ceil((k' + k + 1) / 16) at k = 10

// Number of workgroups the fused pass wants for n rows on a device with n_cu CUs.
int fused_grid(int64_t n, int n_cu);
size_t fused_lds_bytes(int k);
size_t fused_qtab_bytes();  // device scratch the fused launchers need in PassArgs::qtab / qscale / qflag
void fused_qtab_layout(void *base, PassArgs &a);  // points the three at a buffer of fused_qtab_bytes()
// Runs only the slice-table / guard kernel of a model (what every fused pass launches first).  *forced = 0 / 1 when
// the environment pins the engine (int8 without guard / fp64), -1 when qflag decides.
hipError_t launch_gram_guard(int k, const PassArgs &a, hipStream_t s, int *forced);
int fused_gram_tiles(int k);  // number of qflag entries the guard writes for state size k
// Launchers.  Return hipSuccess or the launch error.
hipError_t launch_pass_em(int k, int grid, const PassArgs &a, hipStream_t s);
hipError_t launch_pass_post(int k, int grid, const PassArgs &a, hipStream_t s);
// The EM pass as an eight-wave workgroup, two roles (ppca_em8.hip): front waves stage / [G | b] / solve / cross, back
// waves contract the mask-side statistics on the int8 MFMA.  Honours a.qflag like the int8 instantiation of pass_kernel.
bool em8_covers(int k);
hipError_t launch_em8(int k, int grid, const PassArgs &a, hipStream_t s);
// The same with the per-sample solve pipelined across tiles (ppca_em9.hip; PPCA_EM9=1).
bool em9_covers(int k);
hipError_t launch_em9(int k, int grid, const PassArgs &a, hipStream_t s);
hipError_t em9_debug_counters(unsigned long long *out4, int reset, hipStream_t s);
// The EM pass for 11 <= k <= 16, d <= 256 as two fused kernels (ppca_em16.hip): E-step sweep over X writing the rows
// [wP | wz | w] and the tiles' sample masks, then the mask-side statistics on the int8 MFMA.  Both write disjoint parts of
// part[grid][stats_len] (reduce with launch_reduce_partials).
struct Em16Launch {
    const double *X;
    int64_t ldx;
    const double *w;       // nullable
    int64_t n;
    int d;
    const double *model;
    double *part;          // [grid][stats_len]
    signed char *qtab;     // int8 Gram slice table of the model, em16_qtab_bytes(k) (launch_qprep16)
    double *qscale;        // 16 x tiles dequantisation multipliers
    int *qflag;            // [tiles] guard flags; any set -> the Gram rows are read from Gext
    const double *Gext;    // [n][k'] packed Gram rows of the fp64 engine (read only when the guard tripped)
    double *Wrows;         // [n][em16_ncol(k)] scratch: the rows handed from the first kernel to the second
    unsigned *Mb;          // [ceil(n / 32)][256] scratch: per tile and dimension, the 32 sample bits
    int no_llk;
    double *dbg;
};
// diagnostic counters of the int8 statistics contraction (see ppca_em8.hip): read (and optionally reset) on the current device
hipError_t em8_debug_counters(unsigned long long *out4, int reset, hipStream_t s);
hipError_t em16_debug_counters(unsigned long long *out4, int reset, hipStream_t s);
bool em16_covers(int d, int k);
int em16_ncol(int k);
size_t em16_qtab_bytes(int k);
hipError_t launch_qprep16(int k, const double *model, int d, double *qscale, signed char *qtab, int *qflag, hipStream_t s);
hipError_t launch_em16(int k, int grid, const Em16Launch &a, hipStream_t s);
// The log-likelihood sweep alone (ppca_llk.hip): per-sample llks (nullable) and the per-workgroup scalars; honours
// a.qflag like the int8 instantiation of pass_kernel.
hipError_t launch_llk2(int k, int grid, const PassArgs &a, hipStream_t s);
// PPCAModel::smooth / extrapolate (recon modes 0 / 1) on the same eight-wave sweep (ppca_llk.hip, llk8_run<K, OUT>)
bool recon8_covers(const PassArgs &a);
hipError_t launch_recon8(int k, int grid, const PassArgs &a, hipStream_t s);
// ... of ALL components of a mixture in one launch (same state size k <= FUSED_MAX_K, nm <= MIX_MAX): units = (component, run of
// tiles), dealt so that the workgroups of one XCD walk the same rows for the different components (X from HBM once per iteration)
bool mix_llk8_available();  // false under PPCA_LLK8=0
int mix_llk_runs_per_xcd(int grid, int nm);
hipError_t launch_mix_llk8(int k, int grid, const MixLlkArgs &a, hipStream_t s);
// the fp64-Gram instantiation of the post pass alone, behind a.qflag (the fallback of the int8 llk sweeps: returns at once unless the
// model's table tripped the dynamic-range guard)
hipError_t launch_pass_post_fp64(int k, int grid, const PassArgs &a, hipStream_t s);
int fused_gram_mode();  // 0: int8 behind the guard (default); 1: fp64 pinned (PPCA_GRAM_FP64=1); 2: int8 without guard (tuning builds)
hipError_t launch_qprep_multi(int k, const MixTabArgs &a, hipStream_t s);
hipError_t launch_finalize_qprep_multi(int k, const MixFinalArgs &a, hipStream_t s);
hipError_t launch_reduce_wguard_multi(int k, const MixReduceArgs &a, int grid_parts, hipStream_t s);
hipError_t launch_reduce_partials(const double *part, int grid_parts, int64_t len, double *out, hipStream_t s, int accumulate = 0,
                                  const int *run_if = nullptr);  // run_if: device flag; the kernel returns at once when it is 0
// A guarded EM pass of the fused path after launch_pass_em, in two launches (round 5; rounds 3-4: reduction, wguard_kernel,
// fp64 pass, second reduction):
//   launch_reduce_wguard   sums the partials into `stats` (the fixed order of launch_reduce_partials); the LAST workgroup of
//                          that reduction then decides on the device what the second stage does (GuardArgs, QF_MODE): nothing;
//                          the whole pass again on the fp64 engine (the model tripped the Gram guard, or too many workgroups are
//                          flagged); or only the slices of the workgroups whose cut dominates the rounding bound of a column whose
//                          reduced diagonal entry of S is not large against it -- an outlier row costs its workgroup's slice,
//                          spread over the whole grid, not the pass.
//   launch_em_fallback     the fp64 instantiation of the pass behind that mode (returns at once on 0) into `part2`, then
//                          stats = sum of the un-flagged workgroups' partials + sum of part2 (returns at once on 0).
// Both return hipSuccess without launching anything when the stage does not apply (an engine pinned by the environment,
// kernel-tuning builds): *applies_out tells.
hipError_t launch_reduce_wguard(int k, const double *part, int64_t len, double *stats, const GuardArgs &g, hipStream_t s, bool *applies_out);
hipError_t launch_em_fallback(int k, int grid, PassArgs a, const GuardArgs &g, const double *part, double *part2, int64_t len, double *stats,
                              hipStream_t s);
// finalize_kernel + the slice table / guard flags / padded C of the NEW model in one launch (the plain EM step: the next pass of
// that model then runs with PassArgs::skip_qprep)
hipError_t launch_finalize_qprep(int k, int d, const double *stats, const double *model_in, double *model_out, double tau, int has_ig,
                                 double alpha, double beta, const PassArgs &tab, hipStream_t s);
hipError_t launch_finalize(int k, int d, const double *stats, const double *model_in, double *model_out, double tau,
                           int has_ig, double alpha, double beta, hipStream_t s);
hipError_t launch_synth(const double *c_dev, const double *mean_dev, double *z_work, double *x_out, int64_t row_offset,
                        int64_t n_rows, int d, int k, double sigma, double mask_prob, int mask_kind, int mask_run,
                        uint64_t seed, hipStream_t s);
hipError_t launch_column_presence(const double *X, int64_t ldx, int64_t n, int d, int *present, hipStream_t s);
hipError_t launch_fill(double *p, int64_t n, double v, hipStream_t s);
// dst[i] = finite(src[i]) ? src[i] : NaN (src, dst 16-byte aligned)
hipError_t launch_canon_copy(const double *src, double *dst, int64_t n, hipStream_t s);
hipError_t launch_scale_rows(double *X, int64_t ldx, int d, const int64_t *rows_dev, int64_t n_rows, double factor, hipStream_t s);
// debug: C/D layout probe of v_mfma_f64_16x16x4_f64 (out: 16 x 16 row-major)
hipError_t launch_mfma_i8_probe(const int *a, const int *b, int *out, hipStream_t s);
hipError_t launch_mfma_probe(const double *a16x4, const double *b4x16, double *out16x16, hipStream_t s);

// generic split pipeline (ppca_generic.hip): any d, k <= GENERIC_MAX_K (k <= 64 on the tuned kernels; 65..128 on fp64 contractions and a
// workgroup-per-matrix solver whose k x k matrix is what 160 KB of LDS hold: correct, no performance claim)
constexpr int GENERIC_MAX_K = 128;
size_t generic_workspace_bytes(int d, int k, int64_t n);
hipError_t generic_em_accumulate(const double *X, int64_t ldx, const double *w, int64_t n, int d, int k,
                                 const double *model, double *stats, void *ws, int n_cu, hipStream_t s);
hipError_t generic_post(const double *X, int64_t ldx, const double *w, int64_t n, int d, int k, const double *model,
                        double *scal8, double *llks, double *states, double *covs, double *recon, int recon_mode,
                        void *ws, int n_cu, hipStream_t s);
hipError_t generic_gram_guard(int d, int k, const double *model, void *ws, hipStream_t s, const int **flag_dev, int *forced);
hipError_t generic_finalize(int k, int d, const double *stats, const double *model_in, double *model_out, double tau,
                            int has_ig, double alpha, double beta, int n_cu, hipStream_t s);

// mixture helpers
// llk: [n_models][n]; logw: [n_models]; w nullable.  Writes u: [n_models][n] =
// ln w_i + log posterior_ic (-inf when w_i <= 0) and lse[n] (mixture llk per sample).
hipError_t launch_mix_accumulate(double *out, const double *a, const double *dev, const double *mean,
                                 const double *logpost, int c, int nm, int64_t n, int d, int first, hipStream_t s);
hipError_t launch_mix_posteriors(const double *llk, const double *logw_dev, const double *w, int64_t n, int nm,
                                 double *u, double *lse, double *logpost, hipStream_t s);
hipError_t launch_reduce_max(const double *v, int64_t n, double *out_scalar, double *work, hipStream_t s);
// out[i] = exp(v[i] - *max_dev)
hipError_t launch_exp_shift(const double *v, const double *max_dev, int64_t n, double *out, hipStream_t s);
hipError_t launch_reduce_sum(const double *v, const double *w, int64_t n, double *out_scalar, double *work,
                             hipStream_t s, const int *n_dev = nullptr);  // n_dev: the element count lives on the device (n = upper bound)
hipError_t launch_mix_shift(double *mx, int nm, hipStream_t s);  // non-finite maxima -> 0 (mix.rs:312-323), in place
// out[0 .. nm) = log_softmax(ln sums + shift) (mix.rs:324-325, :335), out[nm] = *llk (nullable)
hipError_t launch_mix_logweights(const double *sums, const double *shift, const double *llk, int nm, double *out, hipStream_t s);
// rows[0 .. m) = ascending indices i with exp(v[i] - *shift_dev) > 0, wout[0 .. m) those weights, counts[select_blocks(n)] = m
// (counts: select_blocks(n) + 1 ints of scratch).
hipError_t launch_select_positive(const double *v, const double *shift_dev, int64_t n, int *counts, int *rows, double *wout,
                                  hipStream_t s);
int select_blocks(int64_t n);
// The same for nm components at once (one launch per stage instead of one per component and stage):
//   launch_mix_posteriors2   u, lse as launch_mix_posteriors + per-block partials bpart[nm + 1][select_blocks(n)]: the block maxima of
//                            u_c (NaNs skipped) and the block sums of w_i lse_i; launch_mix_stage2 reduces them in a fixed order into
//                            maxima[nm] and *llk_out.
//   launch_select_multi      counts[nm][nb + 1] (-> exclusive offsets, total at [nb]), rows[nm][n], wout[nm][n] as
//                            launch_select_positive per component, and the sums of the kept weights into sums_out[nm], the row
//                            counts into used_out[nm] (fixed order: per-block sums in wpart[nm][nb], then one workgroup per component).
hipError_t launch_mix_posteriors2(const double *llk, const double *logw_dev, const double *w, int64_t n, int nm, double *u, double *lse,
                                  double *bpart, hipStream_t s);
hipError_t launch_mix_stage2(const double *bpart, int64_t n, int nm, double *maxima, double *llk_out, hipStream_t s);
hipError_t launch_select_multi(const double *u, const double *shift_dev, int64_t n, int nm, int *counts, int *rows, double *wout, double *wpart,
                               double *sums_out, int *used_out, hipStream_t s);

}  // namespace ppca
