/*
 * ppca_hip.h -- C-ABI of the MI355X-native PPCA EM engine (libppca_hip.so).
 *
 * This is the drop-in boundary for ONE path of viodotcom/ppca_rs: the EM hot path
 * (per-sample masked posterior inference + M-step sufficient statistics) and the
 * passes that share its per-sample math (llk / llks / infer / smooth /
 * extrapolate).  The reference has no FFI seam of its own: its only boundary is
 * the PyO3 class surface (src/python_bindings.rs:15-26).  Each entry point below
 * names the reference item it replaces (paths relative to the reference root);
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; opaque handles own device memory
 *   - every function returns 0 (PPCA_OK) or a negative ppca_status; the message of
 *     the last failure on the calling thread is ppca_last_error()
 *   - "host" pointers are caller-owned host memory; "dev" pointers are device
 *     memory on the context's GPU (e.g. a torch tensor's data_ptr())
 *   - matrices are row-major float64; transform C is (d x k), mean is (d)
 *   - a dataset entry is OBSERVED iff it is finite (dataset.rs:19-22); masked
 *     entries keep their non-finite value in device memory and are removed by
 *     selection, never by multiplication (utils.rs:118-127)
 *   - all work is enqueued on the context's stream; entry points that return
 *     host values synchronise that stream, the *_async/_dev ones do not
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     fails with PPCA_ERR_HIP
 */
#ifndef PPCA_HIP_H
#define PPCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPCA_ABI_VERSION 6

typedef enum ppca_status {
    PPCA_OK = 0,
    PPCA_ERR_INVALID = -1,     /* bad argument / shape mismatch (reference: panics, output_covariance.rs:124) */
    PPCA_ERR_HIP = -2,         /* HIP runtime failure or no device */
    PPCA_ERR_UNSUPPORTED = -3, /* (d, k) outside what the kernels cover */
    PPCA_ERR_EMPTY = -4,       /* empty dataset (reference: assert, ppca_model.rs:52; expect :358) */
    PPCA_ERR_NUMERIC = -5      /* e.g. singular mean-prior covariance (prior.rs:40) */
} ppca_status;

typedef struct ppca_ctx ppca_ctx;         /* one GPU + one stream */
typedef struct ppca_dataset ppca_dataset; /* device-resident Dataset (dataset.rs:93-100) */
typedef struct ppca_model ppca_model;     /* device-resident PPCAModel (ppca_model.rs:18-22,40) */

/* Prior (prior.rs:8-15).  Host pointers; mean is (d), mean_covariance is (d x d). */
typedef struct ppca_prior {
    int32_t has_mean_prior;            /* with_mean_prior            prior.rs:32-45 */
    const double *mean;
    const double *mean_covariance;
    int32_t has_isotropic_noise_prior; /* with_isotropic_noise_prior prior.rs:49-56 */
    double isotropic_noise_alpha;
    double isotropic_noise_beta;
    double transformation_precision;   /* with_transformation_precision prior.rs:60-65 */
} ppca_prior;

/* Synthetic data = the model's own generative process (sample_one,
 * ppca_model.rs:164-181): y = C n1 + mean + sigma n2, entry dropped with
 * probability mask_prob (mask_kind 0) or one cyclic run of mask_run dims per
 * sample (mask_kind 1).  Counter-based RNG keyed by (seed, GLOBAL row, column):
 * a shard [row_offset, row_offset + n_rows) of the same seed holds the same rows
 * whatever the number of shards. */
typedef struct ppca_synth_spec {
    int64_t row_offset;
    int64_t n_rows;
    int32_t d;
    int32_t k;
    double sigma;
    double mask_prob;
    int32_t mask_kind;
    int32_t mask_run;
    uint64_t seed;
    const double *transform; /* host (d x k) ground-truth C */
    const double *mean;      /* host (d) */
} ppca_synth_spec;

/* ------------------------------------------------------------------ misc */
const char *ppca_last_error(void);
int32_t ppca_abi_version(void);
/* 1 when (d, k) runs on the fused single-pass kernel, 0 when it runs on the
 * generic split pipeline, negative if unsupported. */
int32_t ppca_path_kind(int32_t d, int32_t k);

/* --------------------------------------------------------------- context */
/* device_id < 0: use the current HIP device.  stream: a hipStream_t to enqueue
 * on (NULL = the library creates its own).  Replaces rayon's process-global pool
 * (rayon parallel iterators, ppca_model.rs:221-227 et al.). */
int ppca_ctx_create(int32_t device_id, void *stream, ppca_ctx **out);
int ppca_ctx_destroy(ppca_ctx *ctx);
int ppca_ctx_set_stream(ppca_ctx *ctx, void *stream);
int ppca_ctx_synchronize(ppca_ctx *ctx);
/* Device blocks released by a context's buffers (output datasets, scratch) are kept
 * for its next allocation of about the same size instead of going through
 * hipFree / hipMalloc (0.2-0.4 s a pair at 8 GB, 30x the kernel that fills them);
 * at most PPCA_POOL_GB GiB (default min(32, an eighth of the device); 0 disables).
 * The cache is invisible to other allocators of the process (torch's): trim before large allocations made elsewhere.
 * ppca_ctx_trim returns the kept blocks to the device now; an allocation hipMalloc
 * refuses for lack of memory does the same for every context before retrying. */
int ppca_ctx_trim(ppca_ctx *ctx, int64_t *released_bytes);
/* When enabled, the library brackets every launch of the dominant EM kernel with
 * HIP events on the context stream; ppca_ctx_kernel_time returns the summed
 * duration and launch count since the last reset (synchronises). */
int ppca_ctx_enable_timing(ppca_ctx *ctx, int32_t enabled);
int ppca_ctx_kernel_time(ppca_ctx *ctx, double *total_ms, int64_t *launches, int32_t reset);

/* --------------------------------------------------------------- dataset */
/* Dataset(ndarray, weights=None)  src/python_bindings.rs:34-64.
 * x: (n x d) float64 with element strides (row_stride, col_stride) in ELEMENTS
 * (numpy views of any layout, :45); weights: n or NULL (= 1.0, dataset.rs:153-158). */
int ppca_dataset_from_host(ppca_ctx *ctx, const double *x, int64_t n, int32_t d, int64_t row_stride,
                           int64_t col_stride, const double *weights, ppca_dataset **out);
/* Borrow device memory (row-major n x d, weights n or NULL); not freed by the library. */
int ppca_dataset_from_device(ppca_ctx *ctx, const double *x_dev, int64_t n, int32_t d, const double *weights_dev,
                             ppca_dataset **out);
/* PPCAModel::sample (ppca_model.rs:186-191) with a seed, generated on device. */
int ppca_dataset_generate(ppca_ctx *ctx, const ppca_synth_spec *spec, ppca_dataset **out);
/* Dataset::with_weights dataset.rs:171-176: shares the rows, new weights (host or device). */
int ppca_dataset_with_weights(ppca_dataset *ds, const double *weights_host, const double *weights_dev,
                              ppca_dataset **out);
/* DatasetChunks src/python_bindings.rs:151-165: rows [start, start+len), shares storage. */
int ppca_dataset_slice(ppca_dataset *ds, int64_t start, int64_t len, ppca_dataset **out);
/* Dataset.concat src/python_bindings.rs:121-133 */
int ppca_dataset_concat(ppca_ctx *ctx, ppca_dataset *const *parts, int32_t n_parts, ppca_dataset **out);
int ppca_dataset_free(ppca_dataset *ds);
int64_t ppca_dataset_len(const ppca_dataset *ds);        /* __len__ :94-96 */
int32_t ppca_dataset_output_size(const ppca_dataset *ds); /* output_size :98-100 */
const double *ppca_dataset_device_x(const ppca_dataset *ds);
const double *ppca_dataset_device_weights(const ppca_dataset *ds);
/* Dataset.numpy :81-92 / masked_vector dataset.rs:64-72: masked entries come back NaN. */
int ppca_dataset_to_host(ppca_dataset *ds, double *out);
int ppca_dataset_weights_to_host(ppca_dataset *ds, double *out); /* weights :106-108 */
/* Dataset::empty_dimensions dataset.rs:194-222: flags[j] = 1 iff dim j is masked in every sample. */
int ppca_dataset_empty_dimensions(ppca_dataset *ds, int32_t *flags);

/* ----------------------------------------------------------------- model */
/* PPCAModel::new ppca_model.rs:43-48 (isotropic_noise = sigma, transform d x k, mean d). */
int ppca_model_create(ppca_ctx *ctx, int32_t d, int32_t k, double sigma, const double *transform,
                      const double *mean, ppca_model **out);
/* An uninitialised model buffer of the given shape (target of ppca_em_finalize). */
int ppca_model_alloc(ppca_ctx *ctx, int32_t d, int32_t k, ppca_model **out);
int ppca_model_download(ppca_model *m, double *sigma, double *transform, double *mean);
int ppca_model_free(ppca_model *m);
int32_t ppca_model_output_size(const ppca_model *m);
int32_t ppca_model_state_size(const ppca_model *m);

/* ------------------------------------------------- EM step (the hot path) */
/* Packed sufficient statistics of one shard (all linear in the sample weights,
 * hence additive across GPUs -- the ONE collective of the path is a sum of this
 * buffer).  k' = k(k+1)/2, symmetric entries lower-packed (e = a(a+1)/2 + b, b <= a):
 *   cross [d*k]   sum_i w_i x~_ij z_i              ppca_model.rs:281-293
 *   S     [d*k']  sum_i w_i m_ij (z_i z_i^T + Sigma_i)   :297-306
 *   U     [d*k]   sum_i w_i m_ij z_i               (for total_deviation, :338-347)
 *   sumx  [d]     sum_i w_i x~_ij
 *   totals[d]     sum_i w_i m_ij                   :348
 *   scalars[8]    square_error (:345), deviations_square_sum (:346),
 *                 llk of the INPUT model (:142-149), sum_i w_i, #non-empty samples, 0, 0, 0
 * with x~ = x - mean on observed dims (0 elsewhere), z_i/Sigma_i the posterior of
 * the INPUT model (infer_one :195-208). */
int64_t ppca_stats_len(int32_t d, int32_t k);

/* E-step + all M-step reductions over the dataset's rows in ONE streaming pass
 * (replaces infer :221-227 + the four sweeps of iterate_with_prior :278-358).
 * stats_dev: device buffer of ppca_stats_len doubles, overwritten.  Asynchronous. */
int ppca_em_accumulate(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *stats_dev);
/* M-step finalisation on device from (all-reduced) statistics: d row solves
 * (:309-322, keep the old row when the system is singular), sigma^2 (:360-371),
 * mean (:373-377); prior hooks :307-308, :367-368, :379-384.  Asynchronous
 * unless a mean prior is given.  out may not alias model_in. */
int ppca_em_finalize(ppca_ctx *ctx, const ppca_model *model_in, const double *stats_dev, const ppca_prior *prior,
                     ppca_model *out);
/* Same arithmetic on host buffers (no GPU needed): used after a CPU-side
 * reduction and by the mean-prior branch. */
int ppca_em_finalize_host(int32_t d, int32_t k, double sigma, const double *transform, const double *mean,
                          const double *stats, const ppca_prior *prior, double *sigma_out, double *transform_out,
                          double *mean_out);
/* PPCAModel::iterate / iterate_with_prior (ppca_model.rs:267-269, :277-393;
 * src/python_bindings.rs:496-507) on a single GPU = accumulate + finalize.
 * llk_in (nullable): log-likelihood of model_in, a by-product of the pass
 * (saves the trainer's extra llk sweep, python/ppca_rs/__init__.py:51).
 * Reading llk_in synchronises; with llk_in == NULL the call is asynchronous. */
int ppca_em_step(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model_in, const ppca_prior *prior,
                 ppca_model *out, double *llk_in);
/* The log-likelihood by-product of the most recent ppca_em_step / ppca_em_step_sharded / ppca_em_step_group on
 * this context (the llk of THAT step's input model, over all shards), for callers that ran the step
 * asynchronously.  Synchronises. */
int ppca_em_last_llk(ppca_ctx *ctx, double *llk);
/* Debug/parity: the packed statistics copied to host. */
int ppca_stats_raw(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *stats_host);

/* ------------------------------------------------ passes sharing the math */
/* PPCAModel::llk :142-149 (weighted total) and llks :152-159 (per sample,
 * unweighted; nullable; host or device destination). */
int ppca_llk(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *total_host, double *per_sample_host);
int ppca_llks_dev(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *per_sample_dev);
/* PPCAModel::infer :221-227 -> InferredMasked.states (n x k) and covariances
 * (n x k x k, nullable)  src/python_bindings.rs:211-234. */
int ppca_infer(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *states_host, double *covs_host);
/* PPCAModel::smooth :237-244 (mode 0) / extrapolate :254-261 (mode 1): a new,
 * fully-unmasked dataset carrying the input weights (:259). */
int ppca_reconstruct(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, int32_t mode, ppca_dataset **out);
/* InferredMasked::smoothed_covariance_diagonal :485-508 (mode 0) and
 * extrapolated_covariance_diagonal :542-577 (mode 1; observed dims -> 0). */
int ppca_covariance_diagonal(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, int32_t mode,
                             ppca_dataset **out);

/* ------------------------------------------- sample-sharded EM across GPUs */
/* The dataset shards by contiguous row blocks (the rule of Dataset.chunks, src/python_bindings.rs:110-118); every
 * statistic above is a weighted sum over samples, so ONE all-reduce(sum) of the packed buffer per iteration
 * replaces the reference's in-process rayon reductions (ppca_model.rs:290-293, :350-358), and every rank finalises
 * the same model.  The collective is RCCL (resolved at run time with dlopen) on the context stream.
 *
 * One process (or thread) per GPU:  rank 0 calls ppca_comm_unique_id, the host ships the 128 bytes to the other
 * ranks by its own means (MPI, a file, torch.distributed's store), every rank calls ppca_comm_create -- a
 * collective call.  One thread driving several GPUs: ppca_comm_create_all + ppca_em_step_group. */
typedef struct ppca_comm ppca_comm;
#define PPCA_UNIQUE_ID_BYTES 128
int ppca_comm_unique_id(void *id_out /* PPCA_UNIQUE_ID_BYTES */);
int ppca_comm_create(ppca_ctx *ctx, int32_t n_ranks, int32_t rank, const void *unique_id, ppca_comm **out);
/* n contexts on n distinct devices of this process -> n communicators (out[n]) of one clique. */
int ppca_comm_create_all(ppca_ctx *const *ctxs, int32_t n, ppca_comm **out);
int ppca_comm_destroy(ppca_comm *comm);
int32_t ppca_comm_n_ranks(const ppca_comm *comm);
int32_t ppca_comm_rank(const ppca_comm *comm);
/* "rccl <version> via <library>" or "unavailable: <why>". */
const char *ppca_comm_backend(void);
/* In-place all-reduce of n doubles of device memory over the ranks, on the context stream (asynchronous).
 * op 0 = sum, 1 = max (the mixture's per-component maxima, mix.rs:312-317). */
int ppca_comm_allreduce(ppca_comm *comm, double *buf_dev, int64_t n, int32_t op);
/* PPCAModel::iterate_with_prior (ppca_model.rs:277-393) over ALL shards: this rank's ppca_em_accumulate, the
 * all-reduce, ppca_em_finalize -- enqueued back to back on the context stream, no host synchronisation unless
 * llk_in (log-likelihood of model_in over all shards) is requested.  A rank whose shard is empty still calls. */
int ppca_em_step_sharded(ppca_comm *comm, ppca_dataset *shard, const ppca_model *model_in, const ppca_prior *prior,
                         ppca_model *out, double *llk_in);
/* The same from ONE host thread for the n communicators of ppca_comm_create_all (shards[i], models_in[i],
 * models_out[i] live on comms[i]'s device; the all-reduces are issued as one RCCL group). */
int ppca_em_step_group(ppca_comm *const *comms, int32_t n, ppca_dataset *const *shards, ppca_model *const *models_in,
                       const ppca_prior *prior, ppca_model *const *models_out, double *llk_in);

/* PPCAMix::iterate_with_prior (mix.rs:281-337) over ALL row shards in ONE call per rank (BASELINE configuration 5):
 * this rank's responsibilities (K log-likelihood sweeps), an all-reduce(MAX) of the K per-component maxima of
 * ln w_i + log r_ic (:312-317 take them over all samples), the K weighted component passes (rows of negligible weight
 * dropped, see ppca_mix_component_stats), ONE all-reduce(SUM) of [K statistic buffers | K weight sums | llk], and the
 * identical finalisation and new log-weights (:324-325, :335) on every rank -- everything enqueued on the context
 * stream, the shifts computed on the device, one synchronisation at the end for the (K + 1) returned values.
 * models_in / models_out: n_models handles on the communicator's device (state sizes may differ per component);
 * llk_in (nullable): mixture log-likelihood of the input over all shards.  A rank whose shard is empty still calls. */
int ppca_mix_em_step_sharded(ppca_comm *comm, ppca_dataset *shard, ppca_model *const *models_in,
                             const double *log_weights_in, int32_t n_models, const ppca_prior *prior,
                             ppca_model *const *models_out, double *log_weights_out, double *llk_in);

/* Rows of this context's shard that each component pass of the most recent ppca_mix_em_step / ppca_mix_em_step_sharded
 * gathered (the others carried a weight below 2^-200 of the component's largest, see ppca_mix_component_stats): what the
 * step EXECUTED, for measurement (bench.py --config 5). */
int ppca_mix_last_rows_used(ppca_ctx *ctx, int64_t *rows, int32_t n_models);

/* ---------------------------------------------------------------- mixture */
/* PPCAMix::iterate_with_prior mix.rs:281-337 on one GPU: per-sample
 * responsibilities (log-softmax of llk_c + log pi_c, :283-295), per-component
 * weighted EM step (:297-330), new log-weights (:335).  models_in/out: n_models
 * handles of one output size d; state sizes may differ per component (mix.rs:50-71).  llk_in (nullable): mixture log-likelihood of the input
 * (PPCAMix::llk :162-174).  Samples with weight <= 0 contribute nothing
 * (documented divergence from :304-309/:326). */
int ppca_mix_em_step(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models_in, const double *log_weights_in,
                     int32_t n_models, const ppca_prior *prior, ppca_model *const *models_out,
                     double *log_weights_out, double *llk_in);
/* PPCAMix::llks :152-159 / llk :162-174 / infer_cluster :179-189 (log posteriors n x n_models). */
int ppca_mix_llk(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, const double *log_weights,
                 int32_t n_models, double *total_host, double *per_sample_host, double *log_posteriors_host);

/* Building blocks of the sample-SHARDED mixture step (one process per GPU; the caller owns the collectives):
 * ppca_mix_responsibilities_dev: u[c][i] = ln w_i + log r_ic (mix.rs:283-309; -inf for w_i <= 0) into u_dev
 *   (n_models x n, component-major) and the per-sample mixture llk (:137-149) into lse_dev (nullable);
 * ppca_vector_max_dev: max_i v_i skipping NaNs (:312-317); ppca_vector_exp_shift_dev: out_i = exp(v_i - shift)
 *   (:320-323) -- the per-component sample weights once the maximum over ALL shards is known;
 * ppca_vector_sum_dev: sum_i v_i (w_i) (:324-325 and the llk total). */
int ppca_mix_responsibilities_dev(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, const double *log_weights,
                                  int32_t n_models, double *u_dev, double *lse_dev);
/* One component of the sharded M-step in one call: weights exp(u_i - shift) (mix.rs:320-323; shift = the maximum over
 * ALL shards), their sum over this shard (nullable), and the component's weighted statistics (ppca_stats_len doubles,
 * device).  Every statistic is linear in the weights, so rows whose weight is below 2^-200 of the component's largest
 * (their terms sit >= 147 binary orders below the fp64 resolution of the sums they would join) are dropped: the pass
 * gathers the others (rows_used, nullable, reports how many).  Synchronises. */
int ppca_mix_component_stats(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, const double *u_dev, double shift,
                             double *stats_dev, double *sum_host, int64_t *rows_used);
int ppca_vector_max_dev(ppca_ctx *ctx, const double *v_dev, int64_t n, double *max_host);
int ppca_vector_exp_shift_dev(ppca_ctx *ctx, const double *v_dev, double shift, int64_t n, double *out_dev);
int ppca_vector_sum_dev(ppca_ctx *ctx, const double *v_dev, const double *w_dev, int64_t n, double *sum_host);

/* PPCAMix::smooth / extrapolate (mix.rs:245-265, through InferredMaskedMix::smoothed :404-412 and
 * ::extrapolated :414-423) and the diagonal covariances around the mixture mean
 * (smoothed_covariance_diagonal :447-461, extrapolated_covariance_diagonal :489-505): posterior-weighted
 * sums over the components.  mode 0 smooth, 1 extrapolate, 2 smoothed covariance diagonal,
 * 3 extrapolated covariance diagonal.  The output dataset carries no weights (the reference collects
 * fresh samples). */
int ppca_mix_reconstruct(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, const double *log_weights,
                         int32_t n_models, int32_t mode, ppca_dataset **out);

/* ------------------------------------------------------------------ debug */
/* Test hook: cap the number of workgroups of every persistent-grid launch of this context (the fused kernels start
 * min(tiles, CUs) workgroups, each walking a contiguous run of 32-sample tiles) so that a dataset small enough for the CPU
 * oracle still gives every workgroup hundreds of tiles -- the steady state of the kernels (tile rings, software
 * pipelines, the periodic flush of the integer accumulators) under the oracle.  n_workgroups <= 0 restores the device's
 * CU count.  Results do not depend on the grid beyond the order of the partial sums. */
int ppca_ctx_set_grid_limit(ppca_ctx *ctx, int32_t n_workgroups);
/* (ABI 6) Test / measurement hook of the EM pass's int8 form of the mask-side statistics (k <= 10): a 32-sample tile with at most
 * max_rows rows that do not fit the fixed-point form -- outlier samples, heavy sample weights -- sends those rows round it (exact
 * fp64 additions into the accumulators, what the reference's f64 sums do with such a row, ppca_model.rs:297-306) instead of raising
 * the exponents of its workgroup for every later row.  Default 8; 0 = the behaviour of ABI <= 5 (every such tile raises the
 * exponents; the device-side guard then repeats the affected workgroups' slices on the fp64 engine: ppca_em_last_fallback), which
 * the tests of that guard select.  Values above 32 mean 32. */
int ppca_ctx_set_heavy_rows(ppca_ctx *ctx, int32_t max_rows);
/* Diagnostic counters of the int8 statistics contraction of the EM pass on this context's device since the last reset:
 * out8[0..3] the eight-wave kernel (k <= 10), out8[4..7] the two-kernel pass (k = 11..16): [0] tiles cut again after the
 * fixed-point exponents were raised (beyond each workgroup's first tile), [1] periodic flushes of the int64
 * accumulators, [2] the largest number of tiles one workgroup walked, [3] launches.  Synchronises. */
int ppca_debug_counters(ppca_ctx *ctx, int64_t *out8, int32_t reset);

/* Which Gram engine the fused passes use for this model: 0 = int8-sliced MFMA with exact integer accumulation,
 * 1 = fp64 MFMA.  Decided on the device per model by a dynamic-range guard (the int8 form keeps 62 bits below each
 * column maximum of vech(c c^T); a model whose rows of C span many orders of magnitude, or whose sigma^2 lies
 * below that resolution, takes the fp64 form; see ppca_kernels.hip, qprep_kernel).  Stands where the reference
 * computes C_o^T C_o in f64 (output_covariance.rs:57-70).  Synchronises. */
int ppca_gram_engine(ppca_ctx *ctx, const ppca_model *model, int32_t *engine);

/* Which guards of the most recent EM pass of the fused path (d <= 256, k <= 10) on this context sent it to the fp64 engine:
 * *gram_unsafe -- the model tripped the dynamic-range guard of the int8-sliced Gram (see ppca_gram_engine);
 * *stats_unsafe -- the reduced statistics were not large against the rounding of the fixed-point form of the mask-side
 * contraction S / U / totals (a sample far above its neighbours, e.g. an outlier row, lifts its workgroup's column
 * exponents; dimensions masked in that sample then sum coarsely cut rows): the pass was repeated with fp64 accumulation,
 * as the reference sums (ppca_model.rs:297-306).  Both decided on the device; this call synchronises. */
int ppca_em_last_guard(ppca_ctx *ctx, int32_t *gram_unsafe, int32_t *stats_unsafe);

/* Bench / test hook: rows[i] of a device-resident dataset multiplied by `factor` in place (masked entries stay masked) -- how
 * `bench.py --outliers` and the guard tests put outlier samples into data generated on the device.  The dataset must own its
 * rows (not a slice or weighted view sharing another's).  Synchronises. */
int ppca_dataset_scale_rows(ppca_dataset *ds, const int64_t *rows, int64_t n_rows, double factor);

/* What the second stage of the most recent EM pass of the fused path did, decided on the device from the guards above
 * (round 5; stands where the reference's sums are plain f64, ppca_model.rs:297-306): *mode 0 = nothing; 1 = the whole pass again
 * on the fp64 engine (the model tripped the Gram guard, or too many workgroups were flagged); 2 = only the slices of the
 * *workgroups workgroups (*rows rows in all) whose fixed-point cut dominated the rounding bound -- an outlier row costs its
 * workgroup's slice, recomputed by the whole grid, not the pass.  *stage_ms (nullable): HIP-event time of the second stages
 * (fallback pass + second reduction) since timing was enabled or this was last called (0 without timing).  Synchronises. */
int ppca_em_last_fallback(ppca_ctx *ctx, int32_t *mode, int32_t *workgroups, int64_t *rows, double *stage_ms);

/* One v_mfma_f64_16x16x4_f64 on host-supplied A (16 x 4) and B (4 x 16), result
 * (16 x 16) written through the C/D lane map the kernels assume (unit test). */
int ppca_debug_mfma_probe(ppca_ctx *ctx, const double *a16x4, const double *b4x16, double *out16x16);

/* One v_mfma_i32_16x16x64_i8 on raw per-lane operand registers (a_regs, b_regs: [64 lanes][16 bytes];
 * out_regs: [64 lanes][4 i32]) -- pins the operand / result lane maps of the int8 Gram path (unit test). */
int ppca_debug_mfma_i8_probe(ppca_ctx *ctx, const int8_t *a_regs, const int8_t *b_regs, int32_t *out_regs);

#ifdef __cplusplus
}
#endif
#endif /* PPCA_HIP_H */
