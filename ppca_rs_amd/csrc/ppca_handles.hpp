// ppca_handles.hpp -- the opaque handles of include/ppca_hip.h and the small host helpers shared by the C-ABI
// sources (ppca_capi.hip, ppca_comm.hip).  Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/ppca_hip.h"
#include "ppca_internal.hpp"

// Block cache of one context.  hipMalloc / hipFree of the multi-GB buffers this library hands out (an N x d output
// dataset is 8 GB at N = 4 M) cost 0.2-0.4 s a pair -- 30x the kernel that fills them -- so blocks released by a
// context's buffers are kept and handed to its next allocation of about the same size.  Reuse is ordered by the
// context's one stream; give() waits for that stream first, which is the guarantee hipFree gave (work still queued
// against the block finishes before anybody else can own it).  Bounded by `limit` bytes (PPCA_POOL_GB, default
// min(32 GiB, an eighth of the device)); emptied by ppca_ctx_trim, by ppca_ctx_destroy, and by any allocation of the
// process that hipMalloc refuses for lack of memory.
struct DevPool {
    std::mutex mu;
    hipStream_t stream = nullptr;
    bool alive = true;
    size_t cached = 0, limit = 0;
    std::multimap<size_t, void *> blocks;  // capacity -> block
    void *take(size_t cap, size_t *real_cap);
    void give(void *p, size_t cap);
    size_t trim();
    void shutdown();
};

struct DevBuf {
    void *p = nullptr;
    bool owned = true;
    int device = 0;
    size_t cap = 0;
    std::shared_ptr<DevPool> pool;  // where the block goes back to (nullptr: hipFree)
    ~DevBuf() {
        if (!p || !owned) return;
        if (pool)
            pool->give(p, cap);
        else
            (void)hipFree(p);
    }
};
typedef std::shared_ptr<DevBuf> BufRef;

struct ppca_ctx {
    mutable std::recursive_mutex mu;  // see USE_CTX
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;         // workgroups a persistent-grid launch gets (= the device's CUs unless ppca_ctx_set_grid_limit capped it)
    int n_cu_device = 256;  // the device's CUs
    std::shared_ptr<DevPool> pool;  // block cache behind dev_alloc while USE_CTX(this) is in scope
    bool timing = false;
    int skip_llk = 0;  // internal: set around the mixture's component EM steps (PassArgs::no_llk)
    int heavy_max = 8;  // PassArgs::heavy_max of this context's EM passes (ppca_ctx_set_heavy_rows)
    std::vector<const int *> guard_words;  // the guard words (PassArgs::qflag) of the context's LAST fused EM pass -- or of the K component
                                           // passes of its last multi-component mixture step: ppca_em_last_guard / _fallback read these
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    BufRef part;  // per-workgroup partial statistics
    size_t part_cap = 0;
    BufRef stats;  // scratch statistics buffer (em_step, em_step_sharded, stats_raw)
    size_t stats_cap = 0;
    int64_t stats_llk_at = -1;  // where the last EM step left the log-likelihood of its input model in `stats`
    BufRef scal;  // post-pass scalars: [grid][8] partials + 8 reduced
    size_t scal_cap = 0;
    BufRef work;  // 2048 doubles for reductions
    BufRef qtab;  // int8 Gram slice table + scales + guard flags of the model being processed
    size_t qtab_cap = 0;
    BufRef errb;  // [grid + 1][W_GUARD_NCOL] rounding bounds of the int8 mask-side statistics and their column sums, then 2 x [grid]
                  // ints: the workgroups whose slices the fp64 fallback recomputes, as flags and as a list (reduce_wguard_kernel)
    size_t errb_cap = 0;
    // which model the slice table / guard flags / padded C behind qtab belong to (base pointer of the table, device buffer and
    // write stamp of the model): a pass of that model launches no qprep_kernel (PassArgs::skip_qprep)
    const void *qtab_base = nullptr, *qtab_model = nullptr;
    uint64_t qtab_stamp = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events2;  // (timing) the second stage of the guarded EM passes: fallback + reduction
    BufRef gws;   // workspace of the generic split pipeline
    size_t gws_cap = 0;
    // mixture scratch, kept across iterations (hipMalloc / hipFree of hundreds of MB per component cost more than
    // the kernels): 0 llk, 1 u, 2 lse, 3 log posteriors, 4 component weights, 5 row list, 6 block counts
    BufRef mix[7];
    size_t mix_cap[7] = {0, 0, 0, 0, 0, 0, 0};
    BufRef mixpack;  // one-call mixture step: [component statistics ... | weight sums | llk] (the all-reduce(SUM) buffer)
    size_t mixpack_cap = 0;
    BufRef mixaux;   // ... its small device-side vectors: maxima / shifts, new log-weights + llk
    size_t mixaux_cap = 0;
    // The multi-component form of the mixture step (round 6; components of one state size on the fused path, <= MIX_MAX of them): one
    // slice-table block per component (mixq: nm x mixq_stride bytes; slot c holds the table of the model content mixq_slots[c] =
    // (device buffer, write stamp): written by qprep_multi_kernel, or by the step's finalisation for the NEXT iteration), the per-block
    // partials of the responsibilities' maxima / llk and of the weight sums (mixred), per-component row lists / weights / counts
    // (mix[4..6] sized nm x) and per-component partial statistics (part sized nm + 1 x).
    BufRef mixq;
    size_t mixq_cap = 0, mixq_stride = 0;
    const void *mixq_base = nullptr;
    std::vector<std::pair<const void *, uint64_t>> mixq_slots;
    BufRef mixred;
    size_t mixred_cap = 0;
    std::vector<int64_t> mix_rows_used;  // rows of this context's shard each component pass of the last mixture step gathered
    // pinned host memory: a small staging area for asynchronous uploads / downloads of a few values, and the two chunk
    // buffers of the pipelined device-to-host copy (ppca_dataset_to_host, ppca_infer); allocated on first use
    void *hstage = nullptr;
    size_t hstage_cap = 0;
    void *pin[2] = {nullptr, nullptr};
    size_t pin_cap = 0;
    BufRef canon[2];  // device-side chunk buffers of the canonicalising copy (non-finite -> NaN)
    size_t canon_cap[2] = {0, 0};
};

struct ppca_dataset {
    ppca_ctx *ctx = nullptr;
    BufRef xbuf, wbuf;
    const double *X = nullptr;
    const double *w = nullptr;  // nullptr = all ones
    int64_t n = 0;
    int d = 0;
};

struct ppca_model {
    ppca_ctx *ctx = nullptr;
    int d = 0, k = 0;  // k: the state size the kernels run with
    // State size 0 (an isotropic Gaussian around the mean; the reference accepts it, ppca_model.rs:51-70, :399-402) is
    // carried as ONE zero transform column: with c_j = 0 every pass reproduces the k = 0 model exactly (G = 0, M = sigma^2,
    // z = 0, Sigma = 1; ln det M + 2 ln(sigma)(m - 1) = 2 m ln(sigma); cross = 0 keeps the column at zero; the trace
    // term sigma^2 (k - sigma^2 tr M^-1) vanishes).  zero_state marks such a model: the caller sees state size 0.
    bool zero_state = false;
    int k_user() const { return zero_state ? 0 : k; }
    BufRef buf;
    double *p() const { return static_cast<double *>(buf->p); }
    uint64_t stamp = 0;  // changes whenever something is enqueued that writes the buffer (ppca_host::touch): with the buffer's address it
                         // names the model's CONTENT for the contexts' cached slice tables
};

ppca_ctx *ppca_comm_context(ppca_comm *comm);  // ppca_comm.hip (internal)

namespace ppca_host {
// Records the message of the failure for ppca_last_error() on this thread and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int dev_alloc(size_t bytes, BufRef *out);
BufRef dev_borrow(const void *p);
int ensure(BufRef &b, size_t &cap, size_t bytes);
int use_device(const ppca_ctx *ctx);
int ensure_hstage(ppca_ctx *ctx, size_t bytes);
void touch(ppca_model *m);  // the model's buffer is (about to be) written: new stamp
// ppca_em_finalize that also builds the new model's slice table in the same launch (the plain EM steps), ppca_capi.hip
int em_finalize_with_table(ppca_ctx *ctx, const ppca_model *model_in, const double *stats_dev, const ppca_prior *prior, ppca_model *out);
// PPCAMix::iterate_with_prior over the context's rows (comm nullable: one row shard of several), ppca_capi.hip
int mix_em_step(ppca_ctx *ctx, ppca_comm *comm, ppca_dataset *ds, ppca_model *const *models_in, const double *log_weights_in,
                int32_t nm, const ppca_prior *prior, ppca_model *const *models_out, double *log_weights_out, double *llk_in);
// dev_alloc draws from (and its buffers return to) the innermost scope's pool on this thread
struct PoolScope {
    std::shared_ptr<DevPool> *prev;
    explicit PoolScope(std::shared_ptr<DevPool> &pool);
    ~PoolScope();
};
}  // namespace ppca_host

#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t _e = (expr);                                                                                    \
        if (_e != hipSuccess) return ppca_host::fail(PPCA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// Every entry point that touches a context's scratch (partials, statistics, scalars, slice table, events) or reads
// results back holds the context's lock for its whole enqueue-and-read sequence: the reference's methods may be
// called from several Python threads at once (src/python_bindings.rs:466-511 release the GIL; ctypes does too).
// Across calls the single stream orders the reuse of the scratch.  Recursive: entry points build on each other.
#define USE_CTX(c)                                            \
    std::lock_guard<std::recursive_mutex> ctx_lock_((c)->mu); \
    ppca_host::PoolScope pool_scope_((c)->pool);              \
    if (int rc_ = ppca_host::use_device(c)) return rc_
