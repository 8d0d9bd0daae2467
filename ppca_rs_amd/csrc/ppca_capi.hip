// ppca_capi.hip -- host side of the C-ABI declared in include/ppca_hip.h.
// Owns device memory behind opaque handles, sequences the kernels of
// ppca_kernels.hip on the context stream and never computes the hot path on the
// CPU: without a HIP device every compute entry point fails with PPCA_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "ppca_handles.hpp"

using namespace ppca;
using namespace ppca_host;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

int ppca_host::fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// ---- block cache (DevPool, ppca_handles.hpp)
static thread_local std::shared_ptr<DevPool> *g_pool = nullptr;
static std::mutex g_pools_mu;
static std::vector<std::weak_ptr<DevPool>> g_pools;  // every live pool of the process, for the out-of-memory path

ppca_host::PoolScope::PoolScope(std::shared_ptr<DevPool> &pool) : prev(g_pool) { g_pool = &pool; }
ppca_host::PoolScope::~PoolScope() { g_pool = prev; }

static size_t pool_round(size_t bytes) {
    const size_t q = bytes >= (size_t(1) << 20) ? (size_t(2) << 20) : 512;
    return (std::max<size_t>(bytes, 8) + q - 1) / q * q;
}
void *DevPool::take(size_t cap, size_t *real_cap) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = blocks.lower_bound(cap);
    if (it == blocks.end() || it->first > cap + cap / 4) return nullptr;
    void *p = it->second;
    *real_cap = it->first;  // the block keeps its own capacity: it goes back to the cache under that size
    cached -= it->first;
    blocks.erase(it);
    return p;
}
void DevPool::give(void *p, size_t cap) {
    std::unique_lock<std::mutex> lk(mu);
    if (!alive || cap > limit) {
        lk.unlock();
        (void)hipFree(p);
        return;
    }
    (void)hipStreamSynchronize(stream);  // what hipFree guaranteed: nothing queued still touches the block
    std::vector<void *> evict;
    while (cached + cap > limit && !blocks.empty()) {  // largest first: the small blocks are the often-reused ones
        auto last = std::prev(blocks.end());
        cached -= last->first;
        evict.push_back(last->second);
        blocks.erase(last);
    }
    blocks.emplace(cap, p);
    cached += cap;
    lk.unlock();
    for (void *q : evict) (void)hipFree(q);
}
size_t DevPool::trim() {
    std::vector<void *> all;
    size_t bytes;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &b : blocks) all.push_back(b.second);
        blocks.clear();
        bytes = cached;
        cached = 0;
    }
    for (void *q : all) (void)hipFree(q);
    return bytes;
}
void DevPool::shutdown() {
    {
        std::lock_guard<std::mutex> lk(mu);
        alive = false;
        stream = nullptr;
    }
    trim();
}
static std::shared_ptr<DevPool> make_pool(hipStream_t stream) {
    auto pool = std::make_shared<DevPool>();
    pool->stream = stream;
    size_t free_b = 0, total_b = 0;
    // (the cache is invisible to other allocators of the process -- torch's among them: a torch out-of-memory does not
    //  empty it; call ppca_ctx_trim before large allocations made elsewhere.  Hence the modest default.)
    size_t limit = size_t(32) << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) limit = std::min(limit, total_b / 8);
    if (const char *e = getenv("PPCA_POOL_GB")) limit = (size_t)(std::max(atof(e), 0.0) * (double)(size_t(1) << 30));
    pool->limit = limit;
    std::lock_guard<std::mutex> lk(g_pools_mu);
    g_pools.erase(std::remove_if(g_pools.begin(), g_pools.end(), [](const std::weak_ptr<DevPool> &w) { return w.expired(); }),
                  g_pools.end());
    g_pools.push_back(pool);
    return pool;
}
static void trim_all_pools() {
    std::vector<std::shared_ptr<DevPool>> live;
    {
        std::lock_guard<std::mutex> lk(g_pools_mu);
        for (auto &w : g_pools)
            if (auto sp = w.lock()) live.push_back(sp);
    }
    for (auto &sp : live) sp->trim();
}

int ppca_host::dev_alloc(size_t bytes, BufRef *out) {
    auto b = std::make_shared<DevBuf>();
    std::shared_ptr<DevPool> pool = g_pool ? *g_pool : nullptr;
    if (pool && pool->limit == 0) pool = nullptr;
    b->cap = pool ? pool_round(bytes) : std::max<size_t>(bytes, 8);
    if (pool) {
        size_t real = b->cap;
        b->p = pool->take(b->cap, &real);
        if (b->p) b->cap = real;
    }
    if (!b->p) {
        hipError_t e = hipMalloc(&b->p, b->cap);
        if (e == hipErrorOutOfMemory) {  // cached blocks of this or another context may be what is in the way
            (void)hipGetLastError();
            trim_all_pools();
            e = hipMalloc(&b->p, b->cap);
        }
        if (e != hipSuccess) {
            b->p = nullptr;
            return fail(PPCA_ERR_HIP, "hipMalloc of %zu bytes failed: %s", b->cap, hipGetErrorString(e));
        }
    }
    b->pool = pool;
    *out = b;
    return PPCA_OK;
}
BufRef ppca_host::dev_borrow(const void *p) {
    auto b = std::make_shared<DevBuf>();
    b->p = const_cast<void *>(p);
    b->owned = false;
    return b;
}
int ppca_host::ensure(BufRef &b, size_t &cap, size_t bytes) {
    if (cap >= bytes && b) return PPCA_OK;
    BufRef nb;
    int rc = dev_alloc(bytes, &nb);
    if (rc) return rc;
    b = nb;
    cap = bytes;
    return PPCA_OK;
}
int ppca_host::use_device(const ppca_ctx *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return PPCA_OK;
}
int ppca_host::ensure_hstage(ppca_ctx *ctx, size_t bytes) {
    if (ctx->hstage && ctx->hstage_cap >= bytes) return PPCA_OK;
    if (ctx->hstage) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        (void)hipHostFree(ctx->hstage);
        ctx->hstage = nullptr;
    }
    const size_t cap = std::max<size_t>(bytes, 8192);
    HIP_TRY(hipHostMalloc(&ctx->hstage, cap, hipHostMallocDefault));
    ctx->hstage_cap = cap;
    return PPCA_OK;
}

// ------------------------------------------------------------------ device -> host, pipelined
// A pageable hipMemcpy of a large block comes back at ~12-25 GB/s (one staging thread, first-touch page faults of the
// destination on that same thread).  Here the block leaves in 64 MB chunks: [canonicalising copy on the device ->]
// asynchronous copy into one of two pinned buffers -> a handful of host threads move the chunk into the caller's
// (pageable) memory while the next chunk is in flight.  canon: non-finite -> NaN on the way (Dataset.numpy,
// src/python_bindings.rs:81-92 / masked_vector dataset.rs:64-72) -- on the device, not in a host loop.
static int d2h_pipelined(ppca_ctx *ctx, double *dst, const double *src, size_t n, bool canon) {
    constexpr size_t CHUNK = size_t(8) << 20;  // doubles: 64 MB
    if (n == 0) return PPCA_OK;
    if (n < (size_t(1) << 19)) {  // small: one plain copy
        if (canon) {
            BufRef tmp;
            if (int rc = dev_alloc(sizeof(double) * n, &tmp)) return rc;
            HIP_TRY(launch_canon_copy(src, static_cast<double *>(tmp->p), (int64_t)n, ctx->stream));
            HIP_TRY(hipMemcpyAsync(dst, tmp->p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            return PPCA_OK;
        }
        HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return PPCA_OK;
    }
    if (!ctx->pin[0]) {
        for (int b = 0; b < 2; ++b) HIP_TRY(hipHostMalloc(&ctx->pin[b], CHUNK * sizeof(double), hipHostMallocDefault));
        ctx->pin_cap = CHUNK * sizeof(double);
    }
    if (canon)
        for (int b = 0; b < 2; ++b)
            if (int rc = ensure(ctx->canon[b], ctx->canon_cap[b], CHUNK * sizeof(double))) return rc;
    const size_t nch = (n + CHUNK - 1) / CHUNK;
    unsigned hw = std::thread::hardware_concurrency();
    int want = (int)std::min<unsigned>(12u, std::max<unsigned>(1u, hw / 2));
    if (const char *e = getenv("PPCA_D2H_THREADS")) want = std::max(1, atoi(e));
    std::atomic<long> ready{0};
    std::atomic<bool> failed{false}, go{false};
    std::atomic<int> nthreads{0};  // how many copy threads exist (set before `go`): the slices of a chunk are cut for that many
    std::unique_ptr<std::atomic<int>[]> done;
    auto chunk_len = [&](size_t c) { return std::min(CHUNK, n - c * CHUNK); };
    std::vector<std::thread> pool;
    // No C++ exception may cross the C-ABI: allocation and thread creation are fenced; with fewer threads than wanted the
    // copy is cut for those that exist, with none it falls back on the plain copy.
    try {
        done.reset(new std::atomic<int>[nch]);
        for (size_t c = 0; c < nch; ++c) done[c].store(0);
        pool.reserve((size_t)want);
        for (int t = 0; t < want; ++t)
            pool.emplace_back([&, t] {
                while (!go.load(std::memory_order_acquire)) {
                    if (failed.load()) return;
                    std::this_thread::yield();
                }
                const int T = nthreads.load(std::memory_order_acquire);
                for (size_t c = 0; c < nch; ++c) {
                    while (ready.load(std::memory_order_acquire) <= (long)c) {
                        if (failed.load()) return;
                        std::this_thread::yield();
                    }
                    const size_t len = chunk_len(c), per = (len + T - 1) / T, a = std::min(len, per * t), b = std::min(len, a + per);
                    if (b > a) std::memcpy(dst + c * CHUNK + a, static_cast<const double *>(ctx->pin[c & 1]) + a, sizeof(double) * (b - a));
                    done[c].fetch_add(1, std::memory_order_release);
                }
            });
    } catch (...) {
    }
    const int T = (int)pool.size();
    if (T == 0 || !done) {  // no helper thread could be started: the plain copy (canonicalised chunk by chunk when asked)
        failed.store(true);
        for (auto &th : pool) th.join();
        for (size_t c = 0; c < nch; ++c) {
            const size_t len = chunk_len(c);
            const double *from = src + c * CHUNK;
            if (canon) {
                double *stg = static_cast<double *>(ctx->canon[0]->p);
                HIP_TRY(launch_canon_copy(from, stg, (int64_t)len, ctx->stream));
                from = stg;
            }
            HIP_TRY(hipMemcpyAsync(dst + c * CHUNK, from, sizeof(double) * len, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        return PPCA_OK;
    }
    nthreads.store(T, std::memory_order_release);
    go.store(true, std::memory_order_release);
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipError_t err = hipSuccess;
    auto issue = [&](size_t c) -> hipError_t {
        const size_t len = chunk_len(c);
        const double *from = src + c * CHUNK;
        if (canon) {
            double *stg = static_cast<double *>(ctx->canon[c & 1]->p);
            if (hipError_t e = launch_canon_copy(from, stg, (int64_t)len, ctx->stream); e != hipSuccess) return e;
            from = stg;
        }
        if (hipError_t e = hipMemcpyAsync(ctx->pin[c & 1], from, sizeof(double) * len, hipMemcpyDeviceToHost, ctx->stream); e != hipSuccess) return e;
        return hipEventRecord(ev[c & 1], ctx->stream);
    };
    for (int b = 0; b < 2 && err == hipSuccess; ++b) err = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
    if (err == hipSuccess) err = issue(0);
    for (size_t c = 0; c < nch && err == hipSuccess; ++c) {
        if (c + 1 < nch) {
            if (c >= 1)  // the other pinned buffer is free once every thread has emptied chunk c - 1
                while (done[c - 1].load(std::memory_order_acquire) < T) std::this_thread::yield();
            err = issue(c + 1);
            if (err != hipSuccess) break;
        }
        err = hipEventSynchronize(ev[c & 1]);
        if (err == hipSuccess) ready.store((long)c + 1, std::memory_order_release);
    }
    if (err != hipSuccess) failed.store(true);
    for (auto &th : pool) th.join();
    for (int b = 0; b < 2; ++b)
        if (ev[b]) (void)hipEventDestroy(ev[b]);
    if (err != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return fail(PPCA_ERR_HIP, "device-to-host copy failed: %s", hipGetErrorString(err));
    }
    return PPCA_OK;
}

// ------------------------------------------------------------------ misc
extern "C" const char *ppca_last_error(void) { return g_err.c_str(); }
extern "C" int32_t ppca_abi_version(void) { return PPCA_ABI_VERSION; }
extern "C" int32_t ppca_path_kind(int32_t d, int32_t k) {
    if (d < 1 || k < 0) return PPCA_ERR_INVALID;
    if (k == 0) k = 1;  // state size 0 runs as one zero column (ppca_handles.hpp, ppca_model)
    if (d <= FUSED_MAX_D && k <= FUSED_MAX_K) return 1;
    if (k <= GENERIC_MAX_K) return 0;
    return PPCA_ERR_UNSUPPORTED;
}

static int check_path(int d, int k) {
    int kind = ppca_path_kind(d, k);
    if (kind == PPCA_ERR_INVALID) return fail(PPCA_ERR_INVALID, "invalid shape d=%d k=%d", d, k);
    if (kind < 0)
        return fail(PPCA_ERR_UNSUPPORTED, "state size %d is not supported (d=%d): the kernels cover k <= %d", k, d,
                    GENERIC_MAX_K);
    return PPCA_OK;
}

// ------------------------------------------------------------------ context
extern "C" int ppca_ctx_create(int32_t device_id, void *stream, ppca_ctx **out) {
    if (!out) return fail(PPCA_ERR_INVALID, "out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(PPCA_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    int dev = device_id;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= count) return fail(PPCA_ERR_INVALID, "device %d out of range (%d devices)", dev, count);
    HIP_TRY(hipSetDevice(dev));
    auto *ctx = new ppca_ctx();
    ctx->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    ctx->n_cu_device = ctx->n_cu;
    if (const char *e = getenv("PPCA_HEAVY_ROWS")) {  // (A/B runs: the initial value of ppca_ctx_set_heavy_rows)
        const int v = atoi(e);
        ctx->heavy_max = v < 0 ? 0 : (v > 32 ? 32 : v);
    }
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return fail(PPCA_ERR_HIP, "hipStreamCreate failed");
        }
        ctx->own_stream = true;
    }
    ctx->pool = make_pool(ctx->stream);
    size_t cap = 0;
    if (ensure(ctx->work, cap, 2048 * sizeof(double))) {
        delete ctx;
        return PPCA_ERR_HIP;
    }
    *out = ctx;
    return PPCA_OK;
}

extern "C" int ppca_ctx_destroy(ppca_ctx *ctx) {
    if (!ctx) return PPCA_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto *evs : {&ctx->events, &ctx->events2})
        for (auto &ev : *evs) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    for (void *q : ctx->pin)
        if (q) (void)hipHostFree(q);
    if (ctx->pool) ctx->pool->shutdown();  // buffers that outlive the context (datasets, models) fall back to hipFree
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PPCA_OK;
}

extern "C" int ppca_ctx_trim(ppca_ctx *ctx, int64_t *released_bytes) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "ctx is null");
    USE_CTX(ctx);
    const size_t got = ctx->pool ? ctx->pool->trim() : 0;
    if (released_bytes) *released_bytes = (int64_t)got;
    return PPCA_OK;
}

extern "C" int ppca_ctx_set_stream(ppca_ctx *ctx, void *stream) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "ctx is null");
    USE_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) {
        (void)hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    if (ctx->pool) {
        std::lock_guard<std::mutex> lk(ctx->pool->mu);
        ctx->pool->stream = ctx->stream;
    }
    return PPCA_OK;
}

extern "C" int ppca_ctx_synchronize(ppca_ctx *ctx) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "ctx is null");
    USE_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_ctx_enable_timing(ppca_ctx *ctx, int32_t enabled) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "ctx is null");
    std::lock_guard<std::recursive_mutex> ctx_lock_(ctx->mu);
    ctx->timing = enabled != 0;
    return PPCA_OK;
}

extern "C" int ppca_ctx_kernel_time(ppca_ctx *ctx, double *total_ms, int64_t *launches, int32_t reset) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "ctx is null");
    USE_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (auto &ev : ctx->events) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)ctx->events.size();
    if (reset) {
        for (auto &ev : ctx->events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        ctx->events.clear();
    }
    return PPCA_OK;
}

// ------------------------------------------------------------------ dataset
extern "C" int ppca_dataset_from_host(ppca_ctx *ctx, const double *x, int64_t n, int32_t d, int64_t row_stride,
                                      int64_t col_stride, const double *weights, ppca_dataset **out) {
    if (!ctx || !out || n < 0 || d < 1 || (n > 0 && !x)) return fail(PPCA_ERR_INVALID, "bad dataset arguments");
    USE_CTX(ctx);
    auto ds = std::make_unique<ppca_dataset>();
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    if (int rc = dev_alloc(sizeof(double) * (size_t)n * d, &ds->xbuf)) return rc;
    ds->X = static_cast<const double *>(ds->xbuf->p);
    if (n > 0) {
        if (col_stride == 1 && row_stride == d) {
            HIP_TRY(hipMemcpyAsync(ds->xbuf->p, x, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            // arbitrary numpy view (src/python_bindings.rs:45): pack in chunks of rows
            const int64_t chunk = std::max<int64_t>(1, (int64_t)(1 << 22) / d);
            std::vector<double> stage((size_t)chunk * d);
            for (int64_t r0 = 0; r0 < n; r0 += chunk) {
                int64_t rows = std::min(chunk, n - r0);
                for (int64_t r = 0; r < rows; ++r)
                    for (int j = 0; j < d; ++j) stage[(size_t)r * d + j] = x[(r0 + r) * row_stride + j * col_stride];
                HIP_TRY(hipMemcpyAsync(static_cast<double *>(ds->xbuf->p) + r0 * d, stage.data(),
                                       sizeof(double) * (size_t)rows * d, hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
            }
        }
    }
    if (weights) {
        if (int rc = dev_alloc(sizeof(double) * (size_t)n, &ds->wbuf)) return rc;
        if (n > 0) {
            HIP_TRY(hipMemcpyAsync(ds->wbuf->p, weights, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        ds->w = static_cast<const double *>(ds->wbuf->p);
    }
    *out = ds.release();
    return PPCA_OK;
}

extern "C" int ppca_dataset_from_device(ppca_ctx *ctx, const double *x_dev, int64_t n, int32_t d,
                                        const double *weights_dev, ppca_dataset **out) {
    if (!ctx || !out || n < 0 || d < 1 || (n > 0 && !x_dev)) return fail(PPCA_ERR_INVALID, "bad dataset arguments");
    auto ds = std::make_unique<ppca_dataset>();
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    ds->xbuf = dev_borrow(x_dev);
    ds->X = x_dev;
    if (weights_dev) {
        ds->wbuf = dev_borrow(weights_dev);
        ds->w = weights_dev;
    }
    *out = ds.release();
    return PPCA_OK;
}

extern "C" int ppca_dataset_generate(ppca_ctx *ctx, const ppca_synth_spec *spec, ppca_dataset **out) {
    if (!ctx || !spec || !out) return fail(PPCA_ERR_INVALID, "null argument");
    if (spec->n_rows < 0 || spec->d < 1 || spec->k < 0 || !spec->transform || !spec->mean)
        return fail(PPCA_ERR_INVALID, "bad synth spec");
    if (spec->mask_kind == 0 && !(spec->mask_prob >= 0.0 && spec->mask_prob <= 1.0))
        return fail(PPCA_ERR_INVALID, "invalid mask probability");  // ppca_model.rs:171
    USE_CTX(ctx);
    const int d = spec->d, k = spec->k;
    const int64_t n = spec->n_rows;
    auto ds = std::make_unique<ppca_dataset>();
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    if (int rc = dev_alloc(sizeof(double) * (size_t)n * d, &ds->xbuf)) return rc;
    ds->X = static_cast<const double *>(ds->xbuf->p);
    BufRef cbuf, zbuf;
    if (int rc = dev_alloc(sizeof(double) * ((size_t)d * k + d), &cbuf)) return rc;
    double *cdev = static_cast<double *>(cbuf->p);
    double *mdev = cdev + (size_t)d * k;
    if (k > 0) HIP_TRY(hipMemcpyAsync(cdev, spec->transform, sizeof(double) * (size_t)d * k, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(mdev, spec->mean, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
    // generate in row chunks so the latent workspace stays small
    const int64_t chunk = 1 << 20;
    if (int rc = dev_alloc(sizeof(double) * (size_t)std::min(chunk, std::max<int64_t>(n, 1)) * std::max(k, 1), &zbuf)) return rc;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        int64_t rows = std::min(chunk, n - r0);
        HIP_TRY(launch_synth(cdev, mdev, static_cast<double *>(zbuf->p), static_cast<double *>(ds->xbuf->p) + r0 * d,
                             spec->row_offset + r0, rows, d, k, spec->sigma, spec->mask_prob, spec->mask_kind,
                             spec->mask_run, spec->seed, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = ds.release();
    return PPCA_OK;
}

extern "C" int ppca_dataset_with_weights(ppca_dataset *ds, const double *weights_host, const double *weights_dev,
                                         ppca_dataset **out) {
    if (!ds || !out) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ds->ctx);
    auto nd = std::make_unique<ppca_dataset>(*ds);
    if (weights_dev) {
        nd->wbuf = dev_borrow(weights_dev);
        nd->w = weights_dev;
    } else if (weights_host) {
        if (int rc = dev_alloc(sizeof(double) * (size_t)ds->n, &nd->wbuf)) return rc;
        if (ds->n > 0) {
            HIP_TRY(hipMemcpyAsync(nd->wbuf->p, weights_host, sizeof(double) * (size_t)ds->n, hipMemcpyHostToDevice,
                                   ds->ctx->stream));
            HIP_TRY(hipStreamSynchronize(ds->ctx->stream));
        }
        nd->w = static_cast<const double *>(nd->wbuf->p);
    } else {
        nd->wbuf.reset();
        nd->w = nullptr;
    }
    *out = nd.release();
    return PPCA_OK;
}

extern "C" int ppca_dataset_slice(ppca_dataset *ds, int64_t start, int64_t len, ppca_dataset **out) {
    if (!ds || !out || start < 0 || len < 0 || start + len > ds->n) return fail(PPCA_ERR_INVALID, "bad slice");
    auto nd = std::make_unique<ppca_dataset>(*ds);
    nd->X = ds->X + start * ds->d;
    nd->w = ds->w ? ds->w + start : nullptr;
    nd->n = len;
    *out = nd.release();
    return PPCA_OK;
}

extern "C" int ppca_dataset_concat(ppca_ctx *ctx, ppca_dataset *const *parts, int32_t n_parts, ppca_dataset **out) {
    if (!ctx || !out || n_parts < 0 || (n_parts > 0 && !parts)) return fail(PPCA_ERR_INVALID, "null argument");
    if (n_parts == 0) return fail(PPCA_ERR_EMPTY, "cannot concatenate an empty list");
    USE_CTX(ctx);
    int64_t n = 0;
    const int d = parts[0]->d;
    for (int i = 0; i < n_parts; ++i) {
        if (!parts[i] || parts[i]->d != d) return fail(PPCA_ERR_INVALID, "datasets have different output sizes");
        n += parts[i]->n;
    }
    auto ds = std::make_unique<ppca_dataset>();
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    if (int rc = dev_alloc(sizeof(double) * (size_t)n * d, &ds->xbuf)) return rc;
    if (int rc = dev_alloc(sizeof(double) * (size_t)n, &ds->wbuf)) return rc;
    ds->X = static_cast<const double *>(ds->xbuf->p);
    ds->w = static_cast<const double *>(ds->wbuf->p);
    int64_t off = 0;
    for (int i = 0; i < n_parts; ++i) {
        const ppca_dataset *p = parts[i];
        if (p->n == 0) continue;
        HIP_TRY(hipMemcpyAsync(static_cast<double *>(ds->xbuf->p) + off * d, p->X, sizeof(double) * (size_t)p->n * d,
                               hipMemcpyDeviceToDevice, ctx->stream));
        if (p->w) {
            HIP_TRY(hipMemcpyAsync(static_cast<double *>(ds->wbuf->p) + off, p->w, sizeof(double) * (size_t)p->n,
                                   hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            HIP_TRY(launch_fill(static_cast<double *>(ds->wbuf->p) + off, p->n, 1.0, ctx->stream));
        }
        off += p->n;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = ds.release();
    return PPCA_OK;
}

extern "C" int ppca_dataset_free(ppca_dataset *ds) {
    if (ds) {
        (void)hipSetDevice(ds->ctx->device);
        delete ds;
    }
    return PPCA_OK;
}
extern "C" int64_t ppca_dataset_len(const ppca_dataset *ds) { return ds ? ds->n : 0; }
extern "C" int32_t ppca_dataset_output_size(const ppca_dataset *ds) { return ds ? ds->d : 0; }
extern "C" const double *ppca_dataset_device_x(const ppca_dataset *ds) { return ds ? ds->X : nullptr; }
extern "C" const double *ppca_dataset_device_weights(const ppca_dataset *ds) { return ds ? ds->w : nullptr; }

extern "C" int ppca_dataset_to_host(ppca_dataset *ds, double *out) {
    if (!ds || (!out && ds->n > 0)) return fail(PPCA_ERR_INVALID, "null argument");
    if (ds->n == 0) return PPCA_OK;
    USE_CTX(ds->ctx);
    // masked_vector (dataset.rs:64-72): masked -> NaN; +-inf inputs are masked, so they come back NaN too
    return d2h_pipelined(ds->ctx, out, ds->X, (size_t)ds->n * ds->d, true);
}

extern "C" int ppca_dataset_weights_to_host(ppca_dataset *ds, double *out) {
    if (!ds || (!out && ds->n > 0)) return fail(PPCA_ERR_INVALID, "null argument");
    if (ds->n == 0) return PPCA_OK;
    if (!ds->w) {
        for (int64_t i = 0; i < ds->n; ++i) out[i] = 1.0;
        return PPCA_OK;
    }
    USE_CTX(ds->ctx);
    HIP_TRY(hipStreamSynchronize(ds->ctx->stream));
    HIP_TRY(hipMemcpy(out, ds->w, sizeof(double) * (size_t)ds->n, hipMemcpyDeviceToHost));
    return PPCA_OK;
}

extern "C" int ppca_dataset_empty_dimensions(ppca_dataset *ds, int32_t *flags) {
    if (!ds || !flags) return fail(PPCA_ERR_INVALID, "null argument");
    ppca_ctx *ctx = ds->ctx;
    USE_CTX(ctx);
    BufRef pres;
    if (int rc = dev_alloc(sizeof(int) * (size_t)ds->d, &pres)) return rc;
    HIP_TRY(hipMemsetAsync(pres->p, 0, sizeof(int) * (size_t)ds->d, ctx->stream));
    HIP_TRY(launch_column_presence(ds->X, ds->d, ds->n, ds->d, static_cast<int *>(pres->p), ctx->stream));
    std::vector<int> h(ds->d);
    HIP_TRY(hipMemcpyAsync(h.data(), pres->p, sizeof(int) * (size_t)ds->d, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < ds->d; ++j) flags[j] = h[j] ? 0 : 1;
    return PPCA_OK;
}

extern "C" int ppca_dataset_scale_rows(ppca_dataset *ds, const int64_t *rows, int64_t n_rows, double factor) {
    if (!ds || (n_rows > 0 && !rows) || n_rows < 0) return fail(PPCA_ERR_INVALID, "bad arguments");
    for (int64_t i = 0; i < n_rows; ++i)
        if (rows[i] < 0 || rows[i] >= ds->n) return fail(PPCA_ERR_INVALID, "row %lld is outside the dataset", (long long)rows[i]);
    // (a dataset made by ppca_dataset_from_device borrows the caller's `const double *` rows: xbuf->p == X there too, so the
    //  pointer test alone let this hook write through memory the caller had handed over read-only -- advisor, round 5)
    if (!ds->xbuf || !ds->xbuf->owned || ds->X != static_cast<const double *>(ds->xbuf->p))
        return fail(PPCA_ERR_INVALID, "the dataset does not own its rows (a slice, a weighted view or borrowed device memory)");
    ppca_ctx *ctx = ds->ctx;
    USE_CTX(ctx);
    if (n_rows == 0) return PPCA_OK;
    BufRef r;
    if (int rc = dev_alloc(sizeof(int64_t) * (size_t)n_rows, &r)) return rc;
    HIP_TRY(hipMemcpyAsync(r->p, rows, sizeof(int64_t) * (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_scale_rows(static_cast<double *>(ds->xbuf->p), ds->d, ds->d, static_cast<const int64_t *>(r->p), n_rows, factor, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

// ------------------------------------------------------------------ model
namespace ppca_host {
void touch(ppca_model *m) {
    static std::atomic<uint64_t> stamp{0};
    m->stamp = ++stamp;
}
}  // namespace ppca_host
extern "C" int ppca_model_alloc(ppca_ctx *ctx, int32_t d, int32_t k, ppca_model **out) {
    if (!ctx || !out || d < 1 || k < 0) return fail(PPCA_ERR_INVALID, "bad model shape");
    USE_CTX(ctx);
    auto m = std::make_unique<ppca_model>();
    m->ctx = ctx;
    m->d = d;
    m->zero_state = k == 0;
    m->k = k == 0 ? 1 : k;
    if (int rc = dev_alloc(sizeof(double) * (size_t)model_len(d, m->k), &m->buf)) return rc;
    if (m->zero_state)  // (the zero column; a finalisation into this model keeps it at zero: cross = 0)
        HIP_TRY(hipMemsetAsync(m->buf->p, 0, sizeof(double) * (size_t)model_len(d, m->k), ctx->stream));
    touch(m.get());
    *out = m.release();
    return PPCA_OK;
}

extern "C" int ppca_model_create(ppca_ctx *ctx, int32_t d, int32_t k, double sigma, const double *transform,
                                 const double *mean, ppca_model **out) {
    if (!mean || (k > 0 && !transform)) return fail(PPCA_ERR_INVALID, "null model arrays");
    ppca_model *m = nullptr;
    if (int rc = ppca_model_alloc(ctx, d, k, &m)) return rc;
    const int ki = m->k;  // (k = 0: one zero column)
    std::vector<double> h((size_t)model_len(d, ki), 0.0);
    h[0] = sigma;
    h[1] = sigma * sigma;  // isotropic_noise.powi(2), output_covariance.rs:62
    h[2] = std::log(sigma);
    h[3] = 0.0;
    if (k > 0) std::memcpy(h.data() + MODEL_HDR, transform, sizeof(double) * (size_t)d * k);
    std::memcpy(h.data() + MODEL_HDR + (size_t)d * ki, mean, sizeof(double) * d);
    touch(m);
    hipError_t e = hipMemcpyAsync(m->p(), h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        delete m;
        return fail(PPCA_ERR_HIP, "model upload failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return PPCA_OK;
}

extern "C" int ppca_model_download(ppca_model *m, double *sigma, double *transform, double *mean) {
    if (!m) return fail(PPCA_ERR_INVALID, "null model");
    USE_CTX(m->ctx);
    std::vector<double> h((size_t)model_len(m->d, m->k));
    HIP_TRY(hipMemcpyAsync(h.data(), m->p(), sizeof(double) * h.size(), hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    if (sigma) *sigma = h[0];
    if (transform && m->k_user() > 0) std::memcpy(transform, h.data() + MODEL_HDR, sizeof(double) * (size_t)m->d * m->k);
    if (mean) std::memcpy(mean, h.data() + MODEL_HDR + (size_t)m->d * m->k, sizeof(double) * m->d);
    return PPCA_OK;
}

extern "C" int ppca_model_free(ppca_model *m) {
    if (m) {
        (void)hipSetDevice(m->ctx->device);
        delete m;
    }
    return PPCA_OK;
}
extern "C" int32_t ppca_model_output_size(const ppca_model *m) { return m ? m->d : 0; }
extern "C" int32_t ppca_model_state_size(const ppca_model *m) { return m ? m->k_user() : 0; }

// ------------------------------------------------------------------ EM step
extern "C" int64_t ppca_stats_len(int32_t d, int32_t k) { return StatsLayout(d, k == 0 ? 1 : k).len; }  // (k = 0: one zero column)

static int check_pair(const ppca_dataset *ds, const ppca_model *model) {
    if (!ds || !model) return fail(PPCA_ERR_INVALID, "null dataset or model");
    if (ds->d != model->d)  // assert_eq!(mask.0.len(), self.output_size()) output_covariance.rs:124
        return fail(PPCA_ERR_INVALID, "dataset has %d dimensions but the model has output size %d", ds->d, model->d);
    if (ds->ctx->device != model->ctx->device) return fail(PPCA_ERR_INVALID, "dataset and model live on different devices");
    return check_path(model->d, model->k);
}

// Points a fused pass at the context's slice table and tells it whether the table is already this model's (written by the
// previous pass of the same model content, or by finalize_qprep_kernel at the end of the EM step that produced it); the pass
// about to be enqueued makes it this model's either way.
static int qtab_layout(ppca_ctx *ctx, PassArgs &a) {
    if (int rc = ensure(ctx->qtab, ctx->qtab_cap, fused_qtab_bytes())) return rc;
    fused_qtab_layout(ctx->qtab->p, a);
    if (ctx->qtab_base != ctx->qtab->p) {  // a new block: the guard words start at zero (the ticket counter of reduce_wguard_kernel)
        HIP_TRY(hipMemsetAsync(a.qflag, 0, sizeof(int) * 16, ctx->stream));
        ctx->qtab_base = ctx->qtab->p;
        ctx->qtab_model = nullptr;
        ctx->guard_words.clear();  // (they pointed into the previous block)
    }
    return PPCA_OK;
}
static int qtab_bind(ppca_ctx *ctx, const ppca_model *model, PassArgs &a) {
    if (int rc = qtab_layout(ctx, a)) return rc;
    static const bool cache = [] {
        const char *e = getenv("PPCA_QPREP_CACHE");  // 0: every pass builds its table (rounds 1-4)
        return !(e && atoi(e) == 0);
    }();
    a.skip_qprep = cache && ctx->qtab_model == model->buf->p && ctx->qtab_stamp == model->stamp;
    // The table becomes this model's only when the pass that builds it has been ENQUEUED (qtab_commit): an early return between the
    // two -- a failed allocation, a pinned engine that launches no qprep -- must not leave the mark on a table nobody built
    // (advisor, round 5).  Until then the context owns no table.
    if (!a.skip_qprep) ctx->qtab_model = nullptr;
    return PPCA_OK;
}
static void qtab_commit(ppca_ctx *ctx, const ppca_model *model, bool built) {
    if (built) {
        ctx->qtab_model = model->buf->p;
        ctx->qtab_stamp = model->stamp;
    }
}

// rows / wsel / nsel: gathered pass over nsel rows of the dataset (fused path only): sample i is row rows[i] with
// weight wsel[i]; rows == nullptr: the whole dataset with its own weights.
static int em_accumulate_impl(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *stats_dev, const int *rows,
                              const double *wsel, int64_t nsel, const int *nsel_dev = nullptr) {
    if (!ctx || !stats_dev) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    USE_CTX(ctx);
    const StatsLayout L(model->d, model->k);
    const int64_t n = rows ? nsel : ds->n;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(double) * (size_t)L.len, ctx->stream));
        return PPCA_OK;
    }
    if (ppca_path_kind(model->d, model->k) == 0) {
        if (rows) return fail(PPCA_ERR_INVALID, "gathered passes run on the fused path only");
        if (int rc = ensure(ctx->gws, ctx->gws_cap, generic_workspace_bytes(model->d, model->k, ds->n))) return rc;
        hipEvent_t g0 = nullptr, g1 = nullptr;
        if (ctx->timing) {  // the whole split pipeline of this pass, as one timed region
            HIP_TRY(hipEventCreate(&g0));
            HIP_TRY(hipEventCreate(&g1));
            HIP_TRY(hipEventRecord(g0, ctx->stream));
        }
        HIP_TRY(generic_em_accumulate(ds->X, ds->d, ds->w, ds->n, ds->d, model->k, model->p(), stats_dev, ctx->gws->p,
                                      ctx->n_cu, ctx->stream));
        if (ctx->timing) {
            HIP_TRY(hipEventRecord(g1, ctx->stream));
            ctx->events.emplace_back(g0, g1);
        }
        return PPCA_OK;
    }
    const int grid = fused_grid(n, ctx->n_cu);
    // partials of the int8 kernel, then those of the fp64 fallback (launch_em_fallback)
    if (int rc = ensure(ctx->part, ctx->part_cap, sizeof(double) * (size_t)grid * L.len * 2)) return rc;
    PassArgs a{};
    a.X = ds->X;
    a.ldx = ds->d;
    a.w = rows ? wsel : ds->w;
    a.rows = rows;
    a.n = n;
    a.n_dev = rows ? nsel_dev : nullptr;  // gathered pass whose row count lives on the device (nsel = upper bound)
    a.d = ds->d;
    a.model = model->p();
    a.part = static_cast<double *>(ctx->part->p);
    a.no_llk = ctx->skip_llk;
    a.heavy_max = ctx->heavy_max;
    if (int rc = qtab_bind(ctx, model, a)) return rc;
    // the workgroups' rounding bounds, their column sums, then two int arrays: flagged workgroups as flags and as a list
    if (int rc = ensure(ctx->errb, ctx->errb_cap, sizeof(double) * ((size_t)grid + 1) * W_GUARD_NCOL + sizeof(int) * 2 * (size_t)grid)) return rc;
    a.errb = static_cast<double *>(ctx->errb->p);
    GuardArgs g{};
    g.errb = a.errb;
    g.es = a.errb + (size_t)grid * W_GUARD_NCOL;
    g.qflag = a.qflag;
    g.wgflag = reinterpret_cast<int *>(g.es + W_GUARD_NCOL);
    g.who = g.wgflag + grid;
    g.n = a.n;
    g.n_dev = a.n_dev;
    g.d = a.d;
    g.grid = grid;
#ifdef PPCA_PHASE_TIMING
    BufRef dbg;  // [grid][16] phase sums of one thread per role, then [grid][8 waves][16] per-wave sums (em8_kernel)
    if (int rc = dev_alloc(sizeof(double) * (size_t)grid * 144, &dbg)) return rc;
    HIP_TRY(hipMemsetAsync(dbg->p, 0, sizeof(double) * (size_t)grid * 144, ctx->stream));
    a.dbg = static_cast<double *>(dbg->p);
#endif
    hipEvent_t e0 = nullptr, e1 = nullptr, f0 = nullptr, f1 = nullptr;
    if (ctx->timing) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, ctx->stream));
    }
    HIP_TRY(launch_pass_em(model->k, grid, a, ctx->stream));
    qtab_commit(ctx, model, fused_gram_mode() != 1);  // (the fp64-pinned engine launches no qprep_kernel)
    ctx->guard_words.assign(1, a.qflag);
    if (ctx->timing) {
        HIP_TRY(hipEventRecord(e1, ctx->stream));
        ctx->events.emplace_back(e0, e1);
    }
    // Reduction + verdict in one launch, then the second stage behind the verdict: nothing (two launches that return at once),
    // the slices of the flagged workgroups on the fp64 engine, or the whole pass again (ppca_kernels.hip, reduce_wguard_kernel)
    bool guarded = false;
    HIP_TRY(launch_reduce_wguard(model->k, a.part, L.len, stats_dev, g, ctx->stream, &guarded));
    if (guarded) {
        if (ctx->timing) {  // (its own event pair: a pass whose guard trips spends its time here)
            HIP_TRY(hipEventCreate(&f0));
            HIP_TRY(hipEventCreate(&f1));
            HIP_TRY(hipEventRecord(f0, ctx->stream));
        }
        HIP_TRY(launch_em_fallback(model->k, grid, a, g, a.part, a.part + (size_t)grid * L.len, L.len, stats_dev, ctx->stream));
        if (ctx->timing) {
            HIP_TRY(hipEventRecord(f1, ctx->stream));
            ctx->events2.emplace_back(f0, f1);
        }
    }
#ifdef PPCA_PHASE_TIMING
    {
        std::vector<double> h((size_t)grid * 144);
        HIP_TRY(hipMemcpyAsync(h.data(), a.dbg, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        {
            const double tl = (double)((n + FUSED_TILE - 1) / FUSED_TILE) / grid;
            // (em9: "P2" = the Gram's last two digit pairs, "b-loop" = b = X~ C, "gram-1" = mask bytes + the first two digit pairs)
            static const char *fn[16] = {"P2", "wait-back", "stores", "barrier", "b-loop", "barrier", "factor", "solve", "columns", "scalars",
                                         "wg-barrier", "P4a", "barrier", "staging", "barrier", "gram-1"};
            static const char *bn[16] = {"wg-barrier", "digitise", "barrier", "contract", "-", "-", "-", "-", "-", "-", "-", "-", "-", "-", "-", "-"};
            for (int w = 0; w < 8; ++w) {
                double tw[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int g = 0; g < grid; ++g)
                    for (int i = 0; i < 16; ++i) tw[i] += h[(size_t)grid * 16 + ((size_t)g * 8 + w) * 16 + i] / grid;
                fprintf(stderr, "[em8 wave %d cycles/tile]", w);
                for (int i = 0; i < (w < 4 ? 16 : 4); ++i) fprintf(stderr, " %s %.0f", w < 4 ? fn[i] : bn[i], tw[i] / tl);
                fprintf(stderr, "\n");
            }
        }
        double t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < grid; ++g)
            for (int i = 0; i < 16; ++i) t[i] += h[(size_t)g * 16 + i] / grid;
        const double tiles = (double)((n + FUSED_TILE - 1) / FUSED_TILE) / grid;
        fprintf(stderr, "[ppca P2 cycles/tile] mask bytes %.0f  digit pairs {7,6},{5,4} %.0f  b loop %.0f  pairs {3,2},{1,0} + stores + barrier %.0f\n",
                t[12] / tiles, t[13] / tiles, t[14] / tiles, t[1] / tiles);
        fprintf(stderr, "[ppca phase cycles/tile] P1 %.0f  P2 %.0f  P3 %.0f (wave 0: factor %.0f, solve %.0f, columns %.0f, scalars %.0f, barrier %.0f)  P4 %.0f (cross+barrier %.0f, mask+staging %.0f, barrier %.0f)  (tiles/WG %.1f)\n",
                (t[0] + t[5]) / tiles, (t[1] + t[12] + t[13] + t[14]) / tiles, (t[2] + t[8] + t[9] + t[10] + t[11]) / tiles, t[8] / tiles, t[9] / tiles,
                t[10] / tiles, t[11] / tiles, t[2] / tiles, (t[3] + t[4] + t[5] + t[6] + t[7] + t[15]) / tiles,
                (t[4] + t[6]) / tiles, t[7] / tiles, t[3] / tiles, tiles);
        fprintf(stderr, "[ppca P4b int8 cycles/tile] staging %.0f  digitise %.0f  contraction + barriers + stores %.0f\n", t[5] / tiles, t[15] / tiles, t[7] / tiles);
        // (the eight-wave kernel writes slots 8..15; the four-wave kernel's slots 0..7 are then its idle instantiation's)
        fprintf(stderr, "[em8 cycles/tile] front: P2 compute %.0f  wait back + stores + barrier %.0f  P3 %.0f  wg barrier + P4a + barrier %.0f  staging + barrier %.0f"
                        " | back: wait at wg barrier %.0f  digitise + barrier %.0f  contraction / flush %.0f\n",
                t[8] / tiles, t[9] / tiles, t[10] / tiles, t[11] / tiles, t[12] / tiles, t[13] / tiles, t[14] / tiles, t[15] / tiles);
    }
#endif
    return PPCA_OK;
}

extern "C" int ppca_em_accumulate(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *stats_dev) {
    return em_accumulate_impl(ctx, ds, model, stats_dev, nullptr, nullptr, 0);
}

// Dense solve A x = b (n x n, row-major) by LU with partial pivoting; false if singular.
static bool lu_solve(std::vector<double> a, int n, std::vector<double> &b) {
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = std::fabs(a[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(a[(size_t)r * n + c]) > best) { best = std::fabs(a[(size_t)r * n + c]); p = r; }
        if (!(best > 0.0)) return false;
        if (p != c) {
            for (int j = 0; j < n; ++j) std::swap(a[(size_t)c * n + j], a[(size_t)p * n + j]);
            std::swap(b[c], b[p]);
        }
        for (int r = c + 1; r < n; ++r) {
            double f = a[(size_t)r * n + c] / a[(size_t)c * n + c];
            if (f == 0.0) continue;
            for (int j = c; j < n; ++j) a[(size_t)r * n + j] -= f * a[(size_t)c * n + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int j = r + 1; j < n; ++j) s -= a[(size_t)r * n + j] * b[j];
        b[r] = s / a[(size_t)r * n + r];
    }
    return true;
}

// Prior::smooth_mean prior.rs:97-110 with precision = diag(totals) / sigma^2 (ppca_model.rs:379-384)
static int smooth_mean_host(const ppca_prior *prior, int d, const double *totals, double s2, double *mean) {
    const size_t dd = (size_t)d * d;
    // prior precision = inverse of the prior covariance (prior.rs:36-41)
    std::vector<double> prec(dd);
    for (int c = 0; c < d; ++c) {
        std::vector<double> e(d, 0.0);
        e[c] = 1.0;
        if (!lu_solve(std::vector<double>(prior->mean_covariance, prior->mean_covariance + dd), d, e))
            return fail(PPCA_ERR_NUMERIC, "mean covariance should be invertible");
        for (int r = 0; r < d; ++r) prec[(size_t)r * d + c] = e[r];
    }
    std::vector<double> tot(prec), num(d);
    for (int i = 0; i < d; ++i) {
        const double pd = totals[i] / s2;
        tot[(size_t)i * d + i] += pd;
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += prec[(size_t)i * d + j] * prior->mean[j];
        num[i] = s + pd * mean[i];
    }
    if (!lu_solve(tot, d, num)) return fail(PPCA_ERR_NUMERIC, "total precision matrix is not invertible");
    std::memcpy(mean, num.data(), sizeof(double) * d);
    return PPCA_OK;
}

static int check_prior(const ppca_prior *prior) {
    if (!prior) return PPCA_OK;
    if (prior->has_isotropic_noise_prior && !(prior->isotropic_noise_alpha >= 0.0 && prior->isotropic_noise_beta >= 0.0))
        return fail(PPCA_ERR_INVALID, "isotropic noise prior needs alpha >= 0 and beta >= 0");  // prior.rs:50-51
    if (!(prior->transformation_precision >= 0.0))
        return fail(PPCA_ERR_INVALID, "transformation precision must be >= 0");  // prior.rs:61
    if (prior->has_mean_prior && (!prior->mean || !prior->mean_covariance))
        return fail(PPCA_ERR_INVALID, "mean prior arrays are null");
    return PPCA_OK;
}

extern "C" int ppca_em_finalize_host(int32_t d, int32_t k, double sigma, const double *transform, const double *mean,
                                     const double *stats, const ppca_prior *prior, double *sigma_out,
                                     double *transform_out, double *mean_out) {
    if (d < 1 || k < 0 || !mean || !stats || !sigma_out || !mean_out || (k > 0 && (!transform || !transform_out)))
        return fail(PPCA_ERR_INVALID, "null argument");
    if (k == 0) {  // state size 0 = one zero column (statistics of ppca_stats_len(d, 0) = ppca_stats_len(d, 1) doubles)
        std::vector<double> c0((size_t)d, 0.0), c1((size_t)d, 0.0);
        return ppca_em_finalize_host(d, 1, sigma, c0.data(), mean, stats, prior, sigma_out, c1.data(), mean_out);
    }
    if (int rc = check_prior(prior)) return rc;
    (void)sigma;
    const StatsLayout L(d, k);
    const double tau = prior ? prior->transformation_precision : 0.0;
    double totsum = 0.0;
    for (int j = 0; j < d; ++j) totsum += stats[L.totals + j];
    // (clamp: see finalize_kernel, ppca_kernels.hip)
    const double sq = stats[L.scalars + SC_SQERR], dv = std::max(stats[L.scalars + SC_DEVSQ], -stats[L.scalars + SC_SQERR]);
    const double s2new = (prior && prior->has_isotropic_noise_prior)
                             ? ((sq + dv) / 2.0 + prior->isotropic_noise_beta) /
                                   (totsum / 2.0 + prior->isotropic_noise_alpha + 1.0)
                             : (sq + dv) / totsum;
    std::vector<double> S((size_t)L.kp + 1), x((size_t)k + 1);
    for (int j = 0; j < d; ++j) {
        for (int e = 0; e < L.kp; ++e) S[e] = stats[L.S + (int64_t)j * L.kp + e];
        for (int a = 0; a < k; ++a) S[tri(a, a)] += tau;
        double cz = 0.0;
        for (int a = 0; a < k; ++a) {
            x[a] = stats[L.cross + (int64_t)j * k + a];
            cz += transform[(int64_t)j * k + a] * stats[L.U + (int64_t)j * k + a];
        }
        if (chol_packed(S.data(), k)) {
            chol_solve_packed(S.data(), k, x.data());
            for (int a = 0; a < k; ++a) transform_out[(int64_t)j * k + a] = x[a];
        } else {
            for (int a = 0; a < k; ++a) transform_out[(int64_t)j * k + a] = transform[(int64_t)j * k + a];
        }
        const double tot = stats[L.totals + j];
        mean_out[j] = (tot > 0.0 ? (stats[L.sumx + j] - cz) / tot : 0.0) + mean[j];
    }
    if (prior && prior->has_mean_prior) {
        if (int rc = smooth_mean_host(prior, d, stats + L.totals, s2new, mean_out)) return rc;
    }
    *sigma_out = std::sqrt(s2new);
    return PPCA_OK;
}

// with_table: the plain EM steps (ppca_em_step, _sharded, _group) -- the finalisation also builds the slice table, guard flags and
// padded C of the NEW model in the same launch (fused path), so that the next pass of `out` on this context starts at once
static int em_finalize_impl(ppca_ctx *ctx, const ppca_model *model_in, const double *stats_dev, const ppca_prior *prior, ppca_model *out,
                            bool with_table) {
    if (!ctx || !model_in || !stats_dev || !out) return fail(PPCA_ERR_INVALID, "null argument");
    if (out == model_in || out->buf == model_in->buf) return fail(PPCA_ERR_INVALID, "out may not alias model_in");
    if (out->d != model_in->d || out->k != model_in->k || out->zero_state != model_in->zero_state)
        return fail(PPCA_ERR_INVALID, "model shapes differ");
    if (int rc = check_path(model_in->d, model_in->k)) return rc;
    if (int rc = check_prior(prior)) return rc;
    USE_CTX(ctx);
    const int d = model_in->d, k = model_in->k;
    const double tau = prior ? prior->transformation_precision : 0.0;
    const int has_ig = prior ? prior->has_isotropic_noise_prior : 0;
    touch(out);  // (the mean prior below changes the mean only: the slice table depends on the transform and sigma^2)
    if (ppca_path_kind(d, k) == 0) {
        HIP_TRY(generic_finalize(k, d, stats_dev, model_in->p(), out->p(), tau, has_ig,
                                 has_ig ? prior->isotropic_noise_alpha : 0.0, has_ig ? prior->isotropic_noise_beta : 0.0,
                                 ctx->n_cu, ctx->stream));
    } else {
        static const bool fuse = [] {
            const char *e = getenv("PPCA_QPREP_CACHE");  // 0: every pass builds its table (rounds 1-4)
            return !(e && atoi(e) == 0);
        }();
        if (with_table && fuse) {
            PassArgs tab{};
            if (int rc = qtab_layout(ctx, tab)) return rc;
            HIP_TRY(launch_finalize_qprep(k, d, stats_dev, model_in->p(), out->p(), tau, has_ig, has_ig ? prior->isotropic_noise_alpha : 0.0,
                                          has_ig ? prior->isotropic_noise_beta : 0.0, tab, ctx->stream));
            ctx->qtab_model = out->buf->p;
            ctx->qtab_stamp = out->stamp;
        } else {
            HIP_TRY(launch_finalize(k, d, stats_dev, model_in->p(), out->p(), tau, has_ig,
                                    has_ig ? prior->isotropic_noise_alpha : 0.0, has_ig ? prior->isotropic_noise_beta : 0.0,
                                    ctx->stream));
        }
    }
    if (prior && prior->has_mean_prior) {
        // rare branch (d x d solve, prior.rs:97-110): host round trip
        const StatsLayout L(d, k);
        std::vector<double> totals(d), hdr(MODEL_HDR), mean(d);
        HIP_TRY(hipMemcpyAsync(totals.data(), stats_dev + L.totals, sizeof(double) * d, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(hdr.data(), out->p(), sizeof(double) * MODEL_HDR, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(mean.data(), out->p() + MODEL_HDR + (size_t)d * k, sizeof(double) * d, hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        // isotropic_noise_sq as used at ppca_model.rs:382 is the un-rounded variance; hdr[1] = sigma*sigma differs by <= 1 ulp
        if (int rc = smooth_mean_host(prior, d, totals.data(), hdr[1], mean.data())) return rc;
        HIP_TRY(hipMemcpyAsync(out->p() + MODEL_HDR + (size_t)d * k, mean.data(), sizeof(double) * d, hipMemcpyHostToDevice,
                               ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PPCA_OK;
}

extern "C" int ppca_em_finalize(ppca_ctx *ctx, const ppca_model *model_in, const double *stats_dev,
                                const ppca_prior *prior, ppca_model *out) {
    // (with the table: a host that composes the step itself -- accumulate, its own all-reduce, finalize -- runs `out` next)
    return em_finalize_impl(ctx, model_in, stats_dev, prior, out, true);
}
namespace ppca_host {
int em_finalize_with_table(ppca_ctx *ctx, const ppca_model *model_in, const double *stats_dev, const ppca_prior *prior, ppca_model *out) {
    return em_finalize_impl(ctx, model_in, stats_dev, prior, out, true);
}
}  // namespace ppca_host

extern "C" int ppca_em_step(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model_in, const ppca_prior *prior,
                            ppca_model *out, double *llk_in) {
    if (!ctx || !out) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model_in)) return rc;
    if (ds->n == 0) return fail(PPCA_ERR_EMPTY, "dataset is empty");
    USE_CTX(ctx);
    const StatsLayout L(model_in->d, model_in->k);
    if (int rc = ensure(ctx->stats, ctx->stats_cap, sizeof(double) * (size_t)L.len)) return rc;
    double *stats = static_cast<double *>(ctx->stats->p);
    if (int rc = ppca_em_accumulate(ctx, ds, model_in, stats)) return rc;
    if (int rc = em_finalize_impl(ctx, model_in, stats, prior, out, true)) return rc;
    ctx->stats_llk_at = L.scalars + SC_LLK;
    if (llk_in) {
        HIP_TRY(hipMemcpyAsync(llk_in, stats + L.scalars + SC_LLK, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PPCA_OK;
}

extern "C" int ppca_em_last_llk(ppca_ctx *ctx, double *llk) {
    if (!ctx || !llk) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    if (ctx->stats_llk_at < 0 || !ctx->stats) return fail(PPCA_ERR_INVALID, "no EM step has run on this context");
    HIP_TRY(hipMemcpyAsync(llk, static_cast<double *>(ctx->stats->p) + ctx->stats_llk_at, sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_stats_raw(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *stats_host) {
    if (!ctx || !stats_host) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    USE_CTX(ctx);
    const StatsLayout L(model->d, model->k);
    if (int rc = ensure(ctx->stats, ctx->stats_cap, sizeof(double) * (size_t)L.len)) return rc;
    double *stats = static_cast<double *>(ctx->stats->p);
    ctx->stats_llk_at = -1;
    if (int rc = ppca_em_accumulate(ctx, ds, model, stats)) return rc;
    HIP_TRY(hipMemcpyAsync(stats_host, stats, sizeof(double) * (size_t)L.len, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

// ------------------------------------------------------------------ post passes
// Runs pass_kernel<K, false>; scalars (if wanted) end up in ctx->scal[grid*8 .. grid*8+8).
static int run_post(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *llks_dev, double *states_dev,
                    double *covs_dev, double *recon_dev, int recon_mode, double **scal_out) {
    USE_CTX(ctx);
    if (ppca_path_kind(model->d, model->k) == 0) {
        if (int rc = ensure(ctx->scal, ctx->scal_cap, sizeof(double) * 16)) return rc;
        if (int rc = ensure(ctx->gws, ctx->gws_cap, generic_workspace_bytes(model->d, model->k, ds->n))) return rc;
        double *scal8 = static_cast<double *>(ctx->scal->p);
        HIP_TRY(generic_post(ds->X, ds->d, ds->w, ds->n, ds->d, model->k, model->p(), scal8, llks_dev, states_dev,
                             covs_dev, recon_dev, recon_mode, ctx->gws->p, ctx->n_cu, ctx->stream));
        if (scal_out) *scal_out = scal8;
        return PPCA_OK;
    }
    const int grid = fused_grid(ds->n, ctx->n_cu);
    if (int rc = ensure(ctx->scal, ctx->scal_cap, sizeof(double) * ((size_t)grid * 8 + 8))) return rc;
    double *scal = static_cast<double *>(ctx->scal->p);
    if (ds->n == 0) {
        HIP_TRY(hipMemsetAsync(scal + (size_t)grid * 8, 0, sizeof(double) * 8, ctx->stream));
    } else {
        PassArgs a{};
        a.X = ds->X;
        a.ldx = ds->d;
        a.w = ds->w;
        a.n = ds->n;
        a.d = ds->d;
        a.model = model->p();
        a.scal_part = scal;
        a.llks = llks_dev;
        a.states = states_dev;
        a.covs = covs_dev;
        a.recon = recon_dev;
        a.recon_mode = recon_mode;
        if (int rc = qtab_bind(ctx, model, a)) return rc;
#ifdef PPCA_PHASE_TIMING
        BufRef dbg;
        if (int rc = dev_alloc(sizeof(double) * (size_t)grid * 16, &dbg)) return rc;
        HIP_TRY(hipMemsetAsync(dbg->p, 0, sizeof(double) * (size_t)grid * 16, ctx->stream));
        a.dbg = static_cast<double *>(dbg->p);
#endif
        HIP_TRY(launch_pass_post(model->k, grid, a, ctx->stream));
        qtab_commit(ctx, model, fused_gram_mode() != 1);
        HIP_TRY(launch_reduce_partials(scal, grid, 8, scal + (size_t)grid * 8, ctx->stream));
#ifdef PPCA_PHASE_TIMING
        if (!states_dev && !covs_dev && !recon_dev) {
            std::vector<double> h((size_t)grid * 16);
            HIP_TRY(hipMemcpyAsync(h.data(), a.dbg, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            double t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int g = 0; g < grid; ++g)
                for (int i = 0; i < 16; ++i) t[i] += h[(size_t)g * 16 + i] / grid;
            const double tiles = (double)((ds->n + FUSED_TILE - 1) / FUSED_TILE) / grid;
            fprintf(stderr, "[llk2 cycles/tile, thread 0] stage+issue %.0f  barriers %.0f  contract: masks+b %.0f, gram %.0f, stores %.0f  solver %.0f  loop %.0f  (tiles/WG %.1f)\n",
                    t[0] / tiles, t[1] / tiles, t[4] / tiles, t[5] / tiles, t[2] / tiles, t[3] / tiles, t[7] / tiles, tiles);
        }
#endif
    }
    if (scal_out) *scal_out = scal + (size_t)grid * 8;
    return PPCA_OK;
}

extern "C" int ppca_llks_dev(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *per_sample_dev) {
    if (!ctx || !per_sample_dev) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    return run_post(ctx, ds, model, per_sample_dev, nullptr, nullptr, nullptr, 0, nullptr);
}

extern "C" int ppca_llk(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *total_host,
                        double *per_sample_host) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    USE_CTX(ctx);
    BufRef l;
    if (per_sample_host)
        if (int rc = dev_alloc(sizeof(double) * (size_t)ds->n, &l)) return rc;
    double *scal = nullptr;
    if (int rc = run_post(ctx, ds, model, l ? static_cast<double *>(l->p) : nullptr, nullptr, nullptr, nullptr, 0, &scal))
        return rc;
    double h[8];
    HIP_TRY(hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    if (per_sample_host && ds->n > 0)
        HIP_TRY(hipMemcpyAsync(per_sample_host, l->p, sizeof(double) * (size_t)ds->n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (total_host) *total_host = h[SC_LLK];
    if (getenv("PPCA_LLK8_TIMING"))  // (meaningful with a -DLLK8_TIMING build of ppca_llk.hip only: cycle sums over the workgroups)
        fprintf(stderr, "[llk8 cycles, sum over workgroups] contractions %.4g  second staging %.4g  hand-off+solver+table %.4g  last barrier %.4g  other barriers %.4g | wave 1: overlapped staging %.4g\n",
                h[SC_SQERR], h[SC_DEVSQ], h[SC_NONEMPTY], h[5], h[6], h[7]);
    return PPCA_OK;
}

extern "C" int ppca_infer(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, double *states_host,
                          double *covs_host) {
    if (!ctx || (!states_host && !(model && model->zero_state))) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    if (model->zero_state) return PPCA_OK;  // n x 0 states, n x 0 x 0 covariances: nothing to write
    USE_CTX(ctx);
    const int k = model->k;
    BufRef st, cv;
    if (int rc = dev_alloc(sizeof(double) * (size_t)ds->n * k, &st)) return rc;
    if (covs_host)
        if (int rc = dev_alloc(sizeof(double) * (size_t)ds->n * k * k, &cv)) return rc;
    if (int rc = run_post(ctx, ds, model, nullptr, static_cast<double *>(st->p), cv ? static_cast<double *>(cv->p) : nullptr,
                          nullptr, 0, nullptr))
        return rc;
    if (ds->n > 0) {
        if (int rc = d2h_pipelined(ctx, states_host, static_cast<const double *>(st->p), (size_t)ds->n * k, false)) return rc;
        if (covs_host)
            if (int rc = d2h_pipelined(ctx, covs_host, static_cast<const double *>(cv->p), (size_t)ds->n * k * k, false)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

static int recon_common(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, int mode, ppca_dataset **out) {
    if (!ctx || !out) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    USE_CTX(ctx);
    auto nd = std::make_unique<ppca_dataset>();
    nd->ctx = ctx;
    nd->n = ds->n;
    nd->d = ds->d;
    nd->wbuf = ds->wbuf;  // weights carried over (ppca_model.rs:242, :259)
    nd->w = ds->w;
    if (int rc = dev_alloc(sizeof(double) * (size_t)ds->n * ds->d, &nd->xbuf)) return rc;
    nd->X = static_cast<const double *>(nd->xbuf->p);
    if (int rc = run_post(ctx, ds, model, nullptr, nullptr, nullptr, static_cast<double *>(nd->xbuf->p), mode, nullptr))
        return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = nd.release();
    return PPCA_OK;
}

extern "C" int ppca_reconstruct(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, int32_t mode,
                                ppca_dataset **out) {
    if (mode != 0 && mode != 1) return fail(PPCA_ERR_INVALID, "mode must be 0 (smooth) or 1 (extrapolate)");
    return recon_common(ctx, ds, model, mode, out);
}

extern "C" int ppca_covariance_diagonal(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, int32_t mode,
                                        ppca_dataset **out) {
    if (mode != 0 && mode != 1) return fail(PPCA_ERR_INVALID, "mode must be 0 (smoothed) or 1 (extrapolated)");
    return recon_common(ctx, ds, model, 2 + mode, out);
}

// ------------------------------------------------------------------ mixture
static int mix_check(ppca_dataset *ds, ppca_model *const *models, int32_t nm) {
    if (!ds || !models || nm < 1) return fail(PPCA_ERR_INVALID, "bad mixture arguments");
    // components may have different state sizes (mix.rs:50-71, state_sizes :91): every pass below runs per
    // component on that component's own instantiation
    for (int c = 0; c < nm; ++c)
        if (int rc = check_pair(ds, models[c])) return rc;
    return PPCA_OK;
}

// The mixture step's small device vectors, sized by the number of components (mix.rs:50-71 has no limit on it):
// [0, P) maxima -> shifts | [P, 2P) new log-weights, llk | [2P, 3P) the normalised log-weights of the llk sweep |
// [3P, ...) one int per component: rows its pass gathered;  P = nm + 1 rounded up to 8 doubles.
struct MixAux {
    double *shift, *out, *logw;
    int *used;
    size_t P;
};
static int mix_aux(ppca_ctx *ctx, int32_t nm, MixAux &a) {
    a.P = ((size_t)nm + 1 + 7) & ~(size_t)7;
    if (int rc = ensure(ctx->mixaux, ctx->mixaux_cap, sizeof(double) * 3 * a.P + sizeof(int) * (size_t)nm)) return rc;
    a.shift = static_cast<double *>(ctx->mixaux->p);
    a.out = a.shift + a.P;
    a.logw = a.shift + 2 * a.P;
    a.used = reinterpret_cast<int *>(a.shift + 3 * a.P);
    return PPCA_OK;
}

// ---- the multi-component form (round 6): components of ONE state size on the fused path, at most MIX_MAX of them, the int8 engine
// behind its guard (PPCA_MIX_MULTI=0: component by component, as rounds 2-5)
static bool mix_multi_ok(const ppca_dataset *ds, ppca_model *const *models, int nm) {
    static const bool on = [] {
        const char *e = getenv("PPCA_MIX_MULTI");
        return !(e && atoi(e) == 0);
    }();
    if (!on || nm < 1 || nm > MIX_MAX || ds->n <= 0 || ds->n >= ((int64_t)1 << 31)) return false;
    if (fused_gram_mode() != 0 || !mix_llk8_available()) return false;
    for (int c = 0; c < nm; ++c)
        if (ppca_path_kind(models[c]->d, models[c]->k) != 1 || models[c]->k != models[0]->k) return false;
    return true;
}
// One slice-table block per component in ctx->mixq; a slot whose (device buffer, write stamp) is not its model's is rebuilt -- all
// stale slots in ONE qprep launch.  tab[c] = the component's block.
static int mix_tables(ppca_ctx *ctx, ppca_model *const *models, int nm, void **tab) {
    const size_t stride = (fused_qtab_bytes() + 255) & ~(size_t)255;
    if (int rc = ensure(ctx->mixq, ctx->mixq_cap, stride * MIX_MAX)) return rc;
    ctx->mixq_stride = stride;
    if (ctx->mixq->p != ctx->mixq_base) {  // a new block: the guard words start at zero (the ticket counters of the reductions)
        HIP_TRY(hipMemsetAsync(ctx->mixq->p, 0, stride * MIX_MAX, ctx->stream));
        ctx->mixq_base = ctx->mixq->p;
        ctx->guard_words.clear();
        ctx->mixq_slots.assign(MIX_MAX, std::make_pair((const void *)nullptr, (uint64_t)0));
    }
    bool stale = false;
    MixTabArgs a{};
    a.d = models[0]->d;
    a.nm = nm;
    for (int c = 0; c < nm; ++c) {
        tab[c] = static_cast<char *>(ctx->mixq->p) + stride * c;
        a.model[c] = models[c]->p();
        a.tab[c] = tab[c];
        if (ctx->mixq_slots[c] != std::make_pair((const void *)models[c]->buf->p, models[c]->stamp)) stale = true;
    }
    if (stale) {
        HIP_TRY(launch_qprep_multi(models[0]->k, a, ctx->stream));
        for (int c = 0; c < nm; ++c) ctx->mixq_slots[c] = std::make_pair((const void *)models[c]->buf->p, models[c]->stamp);
    }
    return PPCA_OK;
}
// The llk sweeps of all components in one launch (X from HBM once: the workgroups of an XCD walk the same rows for the different
// components, ppca_llk.hip), then per component the fp64 instantiation behind its table's guard flags (returns at once unless the
// component's model tripped the dynamic-range guard).
static int mix_llk_sweeps_multi(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, int nm, double *llk, void **tab) {
    const int64_t n = ds->n;
    const int k = models[0]->k;
    const int grid1 = fused_grid(n, ctx->n_cu);
    if (int rc = ensure(ctx->scal, ctx->scal_cap, sizeof(double) * ((size_t)grid1 * 8 + 16))) return rc;
    MixLlkArgs m{};
    m.X = ds->X;
    m.ldx = ds->d;
    m.n = n;
    m.d = ds->d;
    m.nm = nm;
    const int grid = ctx->n_cu;
    m.runs_per_xcd = mix_llk_runs_per_xcd(grid, nm);
    for (int c = 0; c < nm; ++c) {
        m.model[c] = models[c]->p();
        m.tab[c] = tab[c];
        m.llks[c] = llk + (size_t)c * n;
    }
    HIP_TRY(launch_mix_llk8(k, grid, m, ctx->stream));
    for (int c = 0; c < nm; ++c) {
        PassArgs a{};
        a.X = ds->X;
        a.ldx = ds->d;
        a.w = ds->w;
        a.n = n;
        a.d = ds->d;
        a.model = models[c]->p();
        a.scal_part = static_cast<double *>(ctx->scal->p);
        a.llks = llk + (size_t)c * n;
        fused_qtab_layout(tab[c], a);
        a.skip_qprep = 1;
        HIP_TRY(launch_pass_post_fp64(k, grid1, a, ctx->stream));
    }
    return PPCA_OK;
}
// normalised log-weights (PPCAMix::new mix.rs:66-70) into aux.logw, through the context's pinned staging buffer: an asynchronous
// copy (every entry point that uses the staging ends with a synchronisation, so it is free at the next call)
static int mix_upload_logw(ppca_ctx *ctx, const double *log_weights, int nm, const MixAux &aux) {
    if (int rc = ensure_hstage(ctx, sizeof(double) * (size_t)nm)) return rc;
    double *lw = static_cast<double *>(ctx->hstage);
    double mx = log_weights[0];
    for (int c = 0; c < nm; ++c) mx = std::max(mx, log_weights[c]);
    double s = 0.0;
    for (int c = 0; c < nm; ++c) s += std::exp(log_weights[c] - mx);
    for (int c = 0; c < nm; ++c) lw[c] = log_weights[c] - mx - std::log(s);
    HIP_TRY(hipMemcpyAsync(aux.logw, lw, sizeof(double) * nm, hipMemcpyHostToDevice, ctx->stream));
    return PPCA_OK;
}

// llks of every component -> llk[nm][n]; posteriors/lse on device.
static int mix_posteriors(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, const double *log_weights,
                          int32_t nm, BufRef &llk, BufRef &u, BufRef &lse, BufRef *logpost) {
    const int64_t n = ds->n;
    if (int rc = ensure(ctx->mix[0], ctx->mix_cap[0], sizeof(double) * (size_t)nm * n)) return rc;
    if (int rc = ensure(ctx->mix[1], ctx->mix_cap[1], sizeof(double) * (size_t)nm * n)) return rc;
    if (int rc = ensure(ctx->mix[2], ctx->mix_cap[2], sizeof(double) * (size_t)n)) return rc;
    llk = ctx->mix[0];
    u = ctx->mix[1];
    lse = ctx->mix[2];
    if (logpost) {
        if (int rc = ensure(ctx->mix[3], ctx->mix_cap[3], sizeof(double) * (size_t)nm * n)) return rc;
        *logpost = ctx->mix[3];
    }
    if (mix_multi_ok(ds, models, nm)) {
        void *tab[MIX_MAX];
        if (int rc = mix_tables(ctx, models, nm, tab)) return rc;
        if (int rc = mix_llk_sweeps_multi(ctx, ds, models, nm, static_cast<double *>(llk->p), tab)) return rc;
    } else {
        for (int c = 0; c < nm; ++c)
            if (int rc = ppca_llks_dev(ctx, ds, models[c], static_cast<double *>(llk->p) + (size_t)c * n)) return rc;
    }
    MixAux aux;
    if (int rc = mix_aux(ctx, nm, aux)) return rc;
    if (int rc = mix_upload_logw(ctx, log_weights, nm, aux)) return rc;
    HIP_TRY(launch_mix_posteriors(static_cast<double *>(llk->p), aux.logw, ds->w, n, nm, static_cast<double *>(u->p),
                                  static_cast<double *>(lse->p), logpost ? static_cast<double *>((*logpost)->p) : nullptr,
                                  ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_mix_llk(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, const double *log_weights,
                            int32_t n_models, double *total_host, double *per_sample_host, double *log_posteriors_host) {
    if (!ctx || !log_weights) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = mix_check(ds, models, n_models)) return rc;
    USE_CTX(ctx);
    const int64_t n = ds->n;
    if (n == 0) {  // mix.rs:164-166
        if (total_host) *total_host = 0.0;
        return PPCA_OK;
    }
    BufRef llk, u, lse, lp;
    if (int rc = mix_posteriors(ctx, ds, models, log_weights, n_models, llk, u, lse, log_posteriors_host ? &lp : nullptr))
        return rc;
    double *work = static_cast<double *>(ctx->work->p);
    HIP_TRY(launch_reduce_sum(static_cast<double *>(lse->p), ds->w, n, work + 1024, work, ctx->stream));
    double tot = 0.0;
    HIP_TRY(hipMemcpyAsync(&tot, work + 1024, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (per_sample_host)
        HIP_TRY(hipMemcpyAsync(per_sample_host, lse->p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (log_posteriors_host)
        HIP_TRY(hipMemcpyAsync(log_posteriors_host, lp->p, sizeof(double) * (size_t)n * n_models, hipMemcpyDeviceToHost,
                               ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (total_host) *total_host = tot;
    return PPCA_OK;
}

// One component of the mixture M-step on this context's rows, ENQUEUED only (no host synchronisation): sample weights
// exp(u_i - *shift_dev) (mix.rs:320-323), their sum (:324-325) into *sum_dev, and the component's weighted statistics
// into stats_dev.  Every statistic is linear in the weights, so rows whose weight is below 2^-200 of the component's
// largest (SEL_MIN_WEIGHT, ppca_kernels.hip: their terms sit >= 147 binary orders below the fp64 resolution of the sums
// they would join) are dropped from the pass on the fused path: select_* kernels build the ascending row list and its
// length ON THE DEVICE, the pass gathers those rows and reads the count from device memory (PassArgs::n_dev) -- at
// d = 256 a sample keeps weight in about one component, so the K component passes together cost about as much as one.
static int mix_component_enqueue(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, const double *u_dev,
                                 const double *shift_dev, double *stats_dev, double *sum_dev, const int **rows_used_dev) {
    const int64_t n = ds->n;
    double *work = static_cast<double *>(ctx->work->p);
    const StatsLayout L(model->d, model->k);
    if (rows_used_dev) *rows_used_dev = nullptr;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(double) * (size_t)L.len, ctx->stream));
        HIP_TRY(hipMemsetAsync(sum_dev, 0, sizeof(double), ctx->stream));
        return PPCA_OK;
    }
    if (int rc = ensure(ctx->mix[4], ctx->mix_cap[4], sizeof(double) * (size_t)n)) return rc;
    double *w = static_cast<double *>(ctx->mix[4]->p);
    const bool fused = ppca_path_kind(model->d, model->k) == 1 && n < (int64_t)1 << 31;
    struct SkipLlk {  // the component passes' own llk is never read (the mixture llk comes from the llk sweep)
        ppca_ctx *c;
        explicit SkipLlk(ppca_ctx *ctx_) : c(ctx_) { c->skip_llk = 1; }
        ~SkipLlk() { c->skip_llk = 0; }
    } skip(ctx);
    if (fused) {
        const int nb = select_blocks(n);
        if (int e = ensure(ctx->mix[5], ctx->mix_cap[5], sizeof(int) * (size_t)n)) return e;
        if (int e = ensure(ctx->mix[6], ctx->mix_cap[6], sizeof(int) * ((size_t)nb + 1))) return e;
        int *rows = static_cast<int *>(ctx->mix[5]->p), *counts = static_cast<int *>(ctx->mix[6]->p);
        HIP_TRY(launch_select_positive(u_dev, shift_dev, n, counts, rows, w, ctx->stream));
        HIP_TRY(launch_reduce_sum(w, nullptr, n, sum_dev, work, ctx->stream, counts + nb));
        if (rows_used_dev) *rows_used_dev = counts + nb;
        return em_accumulate_impl(ctx, ds, model, stats_dev, rows, w, n, counts + nb);
    }
    HIP_TRY(launch_exp_shift(u_dev, shift_dev, n, w, ctx->stream));
    HIP_TRY(launch_reduce_sum(w, nullptr, n, sum_dev, work, ctx->stream));
    ppca_dataset *wds = nullptr;
    int rc = ppca_dataset_with_weights(ds, nullptr, w, &wds);  // (borrows w: freeing the handle frees nothing)
    if (rc == PPCA_OK) rc = ppca_em_accumulate(ctx, wds, model, stats_dev);
    ppca_dataset_free(wds);
    return rc;
}

extern "C" int ppca_mix_component_stats(ppca_ctx *ctx, ppca_dataset *ds, const ppca_model *model, const double *u_dev,
                                        double shift, double *stats_dev, double *sum_host, int64_t *rows_used) {
    if (!ctx || !u_dev || !stats_dev) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_pair(ds, model)) return rc;
    USE_CTX(ctx);
    if (int rc = ensure_hstage(ctx, 64)) return rc;
    double *work = static_cast<double *>(ctx->work->p);
    double *hs = static_cast<double *>(ctx->hstage);
    hs[0] = shift;
    HIP_TRY(hipMemcpyAsync(work + 1025, hs, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int *used_dev = nullptr;
    if (int rc = mix_component_enqueue(ctx, ds, model, u_dev, work + 1025, stats_dev, work + 1026, &used_dev)) return rc;
    HIP_TRY(hipMemcpyAsync(hs + 1, work + 1026, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    int *hu = reinterpret_cast<int *>(hs + 2);
    *hu = (int)ds->n;
    if (used_dev) HIP_TRY(hipMemcpyAsync(hu, used_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (sum_host) *sum_host = hs[1];
    if (rows_used) *rows_used = *hu;
    return PPCA_OK;
}

// PPCAMix::iterate_with_prior (mix.rs:281-337) over this context's rows -- all of them (comm == nullptr) or one row
// shard of several (comm: the library's RCCL communicator) -- enqueued on the context stream without any host
// synchronisation until the new log-weights and the log-likelihood are read back at the end:
//   K log-likelihood sweeps -> responsibilities u_ic = ln w_i + log r_ic (:283-309) and the per-sample mixture llk
//   K maxima of u_c (:312-317)                      [all-reduce(MAX) of K doubles: the maxima are over ALL samples]
//   shifts on the device (non-finite -> 0); per component: weights exp(u - shift), their sum, row selection, the
//   gathered weighted EM pass (:320-328)            [ONE all-reduce(SUM) of [K statistics | K sums | llk]]
//   K finalisations (identical on every rank); new log-weights log_softmax(ln sum_c + shift_c) (:324-325, :335)
// The step in its multi-component form (mix_multi_ok; no mean prior): every stage that rounds 2-5 launched once per component is ONE
// launch over all of them, except the gathered EM passes themselves and their (idle) fp64 second stages:
//   slice tables of the K models (cached per model content; built by the previous step's finalisation)        [0-1 launch]
//   mix_llk8_kernel: K llk sweeps, X from HBM once; K fp64 instantiations behind the guard flags (idle)      [1 + K]
//   responsibilities + per-block maxima / llk partials; second stage -> K maxima, llk                        [2]
//                                                       [all-reduce(MAX) of K doubles when sharded]; shifts  [1]
//   row selection of the K components: counts, scan, scatter (+ weight partials), weight sums + row counts   [4]
//   K gathered weighted EM passes (each on its own partials, bounds and guard words)                         [K]
//   reduction + verdict of the K passes; K x (fp64 pass + second reduction) behind the verdicts (idle)       [1 + 2 K]
//                                      [ONE all-reduce(SUM) of [K statistics | K sums | llk] when sharded]
//   K finalisations + the K NEW models' slice tables; new log-weights                                        [2]
// = 12 + 4 K launches (44 at K = 8) where round 5 had ~130, no host synchronisation until the results are read back.
static int mix_em_step_multi(ppca_ctx *ctx, ppca_comm *comm, ppca_dataset *ds, ppca_model *const *models_in,
                             const double *log_weights_in, int32_t nm, const ppca_prior *prior, ppca_model *const *models_out,
                             double *log_weights_out, double *llk_in) {
    const int64_t n = ds->n;
    const int d = ds->d, k = models_in[0]->k;
    const StatsLayout L(d, k);
    const int64_t len = L.len, sums_at = (int64_t)nm * len, llk_at = sums_at + nm, total = llk_at + 1;
    const int grid = fused_grid(n, ctx->n_cu), nb = select_blocks(n);
    MixAux ax;
    if (int rc = mix_aux(ctx, nm, ax)) return rc;
    // every device block of the step is taken here, before the first collective (see mix_em_step)
    const size_t errb_doubles = ((size_t)grid + 1) * W_GUARD_NCOL + (size_t)grid;  // bounds, their sums, 2 x grid ints
    if (int rc = ensure(ctx->mixpack, ctx->mixpack_cap, sizeof(double) * (size_t)total)) return rc;
    if (int rc = ensure(ctx->mix[0], ctx->mix_cap[0], sizeof(double) * (size_t)nm * n)) return rc;
    if (int rc = ensure(ctx->mix[1], ctx->mix_cap[1], sizeof(double) * (size_t)nm * n)) return rc;
    if (int rc = ensure(ctx->mix[2], ctx->mix_cap[2], sizeof(double) * (size_t)n)) return rc;
    if (int rc = ensure(ctx->mix[4], ctx->mix_cap[4], sizeof(double) * (size_t)nm * n)) return rc;
    if (int rc = ensure(ctx->mix[5], ctx->mix_cap[5], sizeof(int) * (size_t)nm * n)) return rc;
    if (int rc = ensure(ctx->mix[6], ctx->mix_cap[6], sizeof(int) * (size_t)nm * ((size_t)nb + 1))) return rc;
    if (int rc = ensure(ctx->mixred, ctx->mixred_cap, sizeof(double) * (size_t)(2 * nm + 1) * nb)) return rc;
    if (int rc = ensure(ctx->part, ctx->part_cap, sizeof(double) * (size_t)(nm + 1) * grid * len)) return rc;
    if (int rc = ensure(ctx->errb, ctx->errb_cap, sizeof(double) * errb_doubles * nm)) return rc;
    if (int rc = ensure_hstage(ctx, sizeof(double) * ax.P + sizeof(int) * (size_t)nm)) return rc;
    void *tab[MIX_MAX];
    if (int rc = mix_tables(ctx, models_in, nm, tab)) return rc;
    double *pack = static_cast<double *>(ctx->mixpack->p);
    double *llk = static_cast<double *>(ctx->mix[0]->p), *u = static_cast<double *>(ctx->mix[1]->p), *lse = static_cast<double *>(ctx->mix[2]->p);
    double *wsel = static_cast<double *>(ctx->mix[4]->p);
    int *rows = static_cast<int *>(ctx->mix[5]->p), *counts = static_cast<int *>(ctx->mix[6]->p);
    double *bpart = static_cast<double *>(ctx->mixred->p), *wpart = bpart + (size_t)(nm + 1) * nb;
    double *part = static_cast<double *>(ctx->part->p), *part2 = part + (size_t)nm * grid * len;
    double *aux = ax.shift;  // maxima -> shifts
    // 1. responsibilities
    if (int rc = mix_llk_sweeps_multi(ctx, ds, models_in, nm, llk, tab)) return rc;
    if (int rc = mix_upload_logw(ctx, log_weights_in, nm, ax)) return rc;
    HIP_TRY(launch_mix_posteriors2(llk, ax.logw, ds->w, n, nm, u, lse, bpart, ctx->stream));
    HIP_TRY(launch_mix_stage2(bpart, n, nm, aux, pack + llk_at, ctx->stream));
    if (comm) {
        if (int rc = ppca_comm_allreduce(comm, aux, nm, 1)) return rc;
    }
    HIP_TRY(launch_mix_shift(aux, nm, ctx->stream));
    // 2. the rows each component keeps, their weights, the weight sums
    HIP_TRY(launch_select_multi(u, aux, n, nm, counts, rows, wsel, wpart, pack + sums_at, ax.used, ctx->stream));
    // 3. the gathered component passes
    ctx->stats_llk_at = -1;
    PassArgs pa[MIX_MAX];
    MixReduceArgs r{};
    r.nm = nm;
    r.len = len;
    for (int c = 0; c < nm; ++c) {
        PassArgs &a = pa[c];
        a = PassArgs{};
        a.X = ds->X;
        a.ldx = d;
        a.w = wsel + (size_t)c * n;
        a.rows = rows + (size_t)c * n;
        a.n = n;
        a.n_dev = counts + (size_t)c * (nb + 1) + nb;
        a.d = d;
        a.model = models_in[c]->p();
        a.part = part + (size_t)c * grid * len;
        a.no_llk = 1;  // (the component passes' own llk is never read)
        a.heavy_max = ctx->heavy_max;
        fused_qtab_layout(tab[c], a);
        a.skip_qprep = 1;
        a.errb = static_cast<double *>(ctx->errb->p) + errb_doubles * c;
        GuardArgs &g = r.g[c];
        g.errb = a.errb;
        g.es = a.errb + (size_t)grid * W_GUARD_NCOL;
        g.qflag = a.qflag;
        g.wgflag = reinterpret_cast<int *>(g.es + W_GUARD_NCOL);
        g.who = g.wgflag + grid;
        g.n = a.n;
        g.n_dev = a.n_dev;
        g.d = d;
        g.grid = grid;
        r.part[c] = a.part;
        r.out[c] = pack + (size_t)c * len;
        HIP_TRY(launch_pass_em(k, grid, a, ctx->stream));
    }
    // 4. reductions + verdicts in one launch, then the (normally idle) second stages
    HIP_TRY(launch_reduce_wguard_multi(k, r, grid, ctx->stream));
    ctx->guard_words.clear();
    for (int c = 0; c < nm; ++c) ctx->guard_words.push_back(r.g[c].qflag);
    for (int c = 0; c < nm; ++c)
        HIP_TRY(launch_em_fallback(k, grid, pa[c], r.g[c], r.part[c], part2, len, r.out[c], ctx->stream));
    if (comm) {
        if (int rc = ppca_comm_allreduce(comm, pack, total, 0)) return rc;
    }
    // 5. finalisations + the new models' tables (slot c is models_out[c]'s from here on), new log-weights
    MixFinalArgs f{};
    f.d = d;
    f.nm = nm;
    f.tau = prior ? prior->transformation_precision : 0.0;
    f.has_ig = prior ? prior->has_isotropic_noise_prior : 0;
    f.alpha = f.has_ig ? prior->isotropic_noise_alpha : 0.0;
    f.beta = f.has_ig ? prior->isotropic_noise_beta : 0.0;
    for (int c = 0; c < nm; ++c) {
        touch(models_out[c]);
        f.stats[c] = pack + (size_t)c * len;
        f.min[c] = models_in[c]->p();
        f.mout[c] = models_out[c]->p();
        f.tab[c] = tab[c];
    }
    HIP_TRY(launch_finalize_qprep_multi(k, f, ctx->stream));
    for (int c = 0; c < nm; ++c) ctx->mixq_slots[c] = std::make_pair((const void *)models_out[c]->buf->p, models_out[c]->stamp);
    HIP_TRY(launch_mix_logweights(pack + sums_at, aux, pack + llk_at, nm, ax.out, ctx->stream));
    double *hs = static_cast<double *>(ctx->hstage);
    HIP_TRY(hipMemcpyAsync(hs, ax.out, sizeof(double) * (size_t)(nm + 1), hipMemcpyDeviceToHost, ctx->stream));
    int *hused = reinterpret_cast<int *>(hs + ax.P);
    HIP_TRY(hipMemcpyAsync(hused, ax.used, sizeof(int) * (size_t)nm, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < nm; ++c) log_weights_out[c] = hs[c];
    if (llk_in) *llk_in = hs[nm];
    ctx->mix_rows_used.assign(nm, 0);
    for (int c = 0; c < nm; ++c) ctx->mix_rows_used[c] = (int64_t)hused[c];
    return PPCA_OK;
}

int ppca_host::mix_em_step(ppca_ctx *ctx, ppca_comm *comm, ppca_dataset *ds, ppca_model *const *models_in,
                           const double *log_weights_in, int32_t nm, const ppca_prior *prior, ppca_model *const *models_out,
                           double *log_weights_out, double *llk_in) {
    USE_CTX(ctx);
    const int64_t n = ds->n;
    for (int c = 0; c < nm; ++c) {
        if (!models_out[c]) return fail(PPCA_ERR_INVALID, "null output model");
        if (models_out[c]->d != models_in[c]->d || models_out[c]->k != models_in[c]->k)
            return fail(PPCA_ERR_INVALID, "model shapes differ (component %d)", c);
        if (models_out[c] == models_in[c] || models_out[c]->buf == models_in[c]->buf)
            return fail(PPCA_ERR_INVALID, "out may not alias model_in (component %d)", c);
    }
    if (int rc = check_prior(prior)) return rc;
    if (!(prior && prior->has_mean_prior) && mix_multi_ok(ds, models_in, nm)) {
        bool distinct = true;  // (the K output buffers are written by ONE launch: they must be K different ones)
        for (int c = 0; c < nm && distinct; ++c)
            for (int e = 0; e < nm; ++e)
                if ((e != c && models_out[e]->buf == models_out[c]->buf) || models_out[c]->buf == models_in[e]->buf) distinct = false;
        if (distinct)
            return mix_em_step_multi(ctx, comm, ds, models_in, log_weights_in, nm, prior, models_out, log_weights_out, llk_in);
    }
    // packed buffer: [statistics of component 0 | ... | nm weight sums | llk]; components may differ in state size
    std::vector<int64_t> off(nm + 1, 0);
    for (int c = 0; c < nm; ++c) off[c + 1] = off[c] + StatsLayout(models_in[c]->d, models_in[c]->k).len;
    const int64_t sums_at = off[nm], llk_at = sums_at + nm, total = llk_at + 1;
    if (int rc = ensure(ctx->mixpack, ctx->mixpack_cap, sizeof(double) * (size_t)total)) return rc;
    MixAux ax;
    if (int rc = mix_aux(ctx, nm, ax)) return rc;
    // Every device block the step will ask for is taken HERE, before the first collective: a rank whose allocation fails
    // returns now, while no peer is inside an all-reduce it would never join (the sizes below are the largest any of the
    // K llk sweeps and component passes of this call can request).
    if (n > 0) {
        size_t max_len = 0, gws_bytes = 0;
        bool any_fused = false;
        for (int c = 0; c < nm; ++c) {
            max_len = std::max(max_len, (size_t)StatsLayout(models_in[c]->d, models_in[c]->k).len);
            if (ppca_path_kind(models_in[c]->d, models_in[c]->k) == 1) any_fused = true;
            else gws_bytes = std::max(gws_bytes, generic_workspace_bytes(models_in[c]->d, models_in[c]->k, n));
        }
        const int grid = fused_grid(n, ctx->n_cu);
        if (int rc = ensure(ctx->mix[0], ctx->mix_cap[0], sizeof(double) * (size_t)nm * n)) return rc;
        if (int rc = ensure(ctx->mix[1], ctx->mix_cap[1], sizeof(double) * (size_t)nm * n)) return rc;
        if (int rc = ensure(ctx->mix[2], ctx->mix_cap[2], sizeof(double) * (size_t)n)) return rc;
        if (int rc = ensure(ctx->mix[4], ctx->mix_cap[4], sizeof(double) * (size_t)n)) return rc;
        if (int rc = ensure(ctx->scal, ctx->scal_cap, sizeof(double) * ((size_t)grid * 8 + 16))) return rc;
        if (any_fused && n < (int64_t)1 << 31) {
            if (int rc = ensure(ctx->mix[5], ctx->mix_cap[5], sizeof(int) * (size_t)n)) return rc;
            if (int rc = ensure(ctx->mix[6], ctx->mix_cap[6], sizeof(int) * ((size_t)select_blocks(n) + 1))) return rc;
            // (what em_accumulate_impl will ask for: the partials of both stages, the bounds + workgroup flags)
            if (int rc = ensure(ctx->part, ctx->part_cap, sizeof(double) * (size_t)grid * max_len * 2)) return rc;
            if (int rc = ensure(ctx->qtab, ctx->qtab_cap, fused_qtab_bytes())) return rc;
            if (int rc = ensure(ctx->errb, ctx->errb_cap, sizeof(double) * ((size_t)grid + 1) * W_GUARD_NCOL + sizeof(int) * 2 * (size_t)grid)) return rc;
        }
        if (gws_bytes)
            if (int rc = ensure(ctx->gws, ctx->gws_cap, gws_bytes)) return rc;
    }
    if (int rc = ensure_hstage(ctx, sizeof(double) * ax.P + sizeof(int) * (size_t)nm)) return rc;
    double *pack = static_cast<double *>(ctx->mixpack->p);
    double *aux = ax.shift;  // maxima -> shifts
    double *work = static_cast<double *>(ctx->work->p);
    BufRef llk, u, lse;
    if (n > 0) {
        if (int rc = mix_posteriors(ctx, ds, models_in, log_weights_in, nm, llk, u, lse, nullptr)) return rc;
        HIP_TRY(launch_reduce_sum(static_cast<double *>(lse->p), ds->w, n, pack + llk_at, work, ctx->stream));
    } else {
        HIP_TRY(hipMemsetAsync(pack + llk_at, 0, sizeof(double), ctx->stream));
    }
    const double *ud = n > 0 ? static_cast<const double *>(u->p) : nullptr;
    for (int c = 0; c < nm; ++c) HIP_TRY(launch_reduce_max(ud ? ud + (size_t)c * n : nullptr, n, aux + c, work, ctx->stream));
    if (comm) {
        if (int rc = ppca_comm_allreduce(comm, aux, nm, 1)) return rc;
    }
    HIP_TRY(launch_mix_shift(aux, nm, ctx->stream));
    ctx->stats_llk_at = -1;
    int *used_all = ax.used;  // rows each component pass gathered (diagnostic: ppca_mix_last_rows_used)
    for (int c = 0; c < nm; ++c) {
        const int *used_dev = nullptr;
        if (int rc = mix_component_enqueue(ctx, ds, models_in[c], ud ? ud + (size_t)c * n : nullptr, aux + c, pack + off[c],
                                           pack + sums_at + c, &used_dev))
            return rc;
        if (used_dev) HIP_TRY(hipMemcpyAsync(used_all + c, used_dev, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        else HIP_TRY(hipMemsetAsync(used_all + c, 0xFF, sizeof(int), ctx->stream));  // (-1: the pass took every row)
    }
    if (comm) {
        if (int rc = ppca_comm_allreduce(comm, pack, total, 0)) return rc;
    }
    for (int c = 0; c < nm; ++c)  // (plain finalisation: the context has ONE slice table, the next iteration walks K models)
        if (int rc = em_finalize_impl(ctx, models_in[c], pack + off[c], prior, models_out[c], false)) return rc;
    HIP_TRY(launch_mix_logweights(pack + sums_at, aux, pack + llk_at, nm, ax.out, ctx->stream));
    double *hs = static_cast<double *>(ctx->hstage);
    HIP_TRY(hipMemcpyAsync(hs, ax.out, sizeof(double) * (size_t)(nm + 1), hipMemcpyDeviceToHost, ctx->stream));
    int *hused = reinterpret_cast<int *>(hs + ax.P);
    HIP_TRY(hipMemcpyAsync(hused, used_all, sizeof(int) * (size_t)nm, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < nm; ++c) log_weights_out[c] = hs[c];
    if (llk_in) *llk_in = hs[nm];
    ctx->mix_rows_used.assign(nm, 0);
    for (int c = 0; c < nm; ++c) ctx->mix_rows_used[c] = hused[c] < 0 ? n : (int64_t)hused[c];
    return PPCA_OK;
}

extern "C" int ppca_mix_last_rows_used(ppca_ctx *ctx, int64_t *rows, int32_t n_models) {
    if (!ctx || !rows) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    if ((size_t)n_models != ctx->mix_rows_used.size()) return fail(PPCA_ERR_INVALID, "the last mixture step had %d components", (int)ctx->mix_rows_used.size());
    for (int c = 0; c < n_models; ++c) rows[c] = ctx->mix_rows_used[c];
    return PPCA_OK;
}

extern "C" int ppca_mix_em_step(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models_in,
                                const double *log_weights_in, int32_t n_models, const ppca_prior *prior,
                                ppca_model *const *models_out, double *log_weights_out, double *llk_in) {
    if (!ctx || !log_weights_in || !models_out || !log_weights_out) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = mix_check(ds, models_in, n_models)) return rc;
    if (ds->n == 0) return fail(PPCA_ERR_EMPTY, "dataset is empty");
    return mix_em_step(ctx, nullptr, ds, models_in, log_weights_in, n_models, prior, models_out, log_weights_out, llk_in);
}

extern "C" int ppca_mix_em_step_sharded(ppca_comm *comm, ppca_dataset *shard, ppca_model *const *models_in,
                                        const double *log_weights_in, int32_t n_models, const ppca_prior *prior,
                                        ppca_model *const *models_out, double *log_weights_out, double *llk_in) {
    if (!comm || !log_weights_in || !models_out || !log_weights_out) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = mix_check(shard, models_in, n_models)) return rc;  // (an empty shard still takes part in the collectives)
    ppca_ctx *ctx = ppca_comm_context(comm);
    if (shard->ctx->device != ctx->device) return fail(PPCA_ERR_INVALID, "the shard does not live on the communicator's device");
    return mix_em_step(ctx, comm, shard, models_in, log_weights_in, n_models, prior, models_out, log_weights_out, llk_in);
}

// ---- building blocks of the SHARDED mixture step (one process per GPU; ppca_rs_amd/distributed.py::ShardedMixEM)
extern "C" int ppca_mix_responsibilities_dev(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models,
                                             const double *log_weights, int32_t n_models, double *u_dev, double *lse_dev) {
    if (!ctx || !log_weights || !u_dev) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = mix_check(ds, models, n_models)) return rc;
    USE_CTX(ctx);
    const int64_t n = ds->n;
    if (n == 0) return PPCA_OK;
    BufRef llk, u, lse;
    if (int rc = mix_posteriors(ctx, ds, models, log_weights, n_models, llk, u, lse, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(u_dev, u->p, sizeof(double) * (size_t)n_models * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (lse_dev) HIP_TRY(hipMemcpyAsync(lse_dev, lse->p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_vector_max_dev(ppca_ctx *ctx, const double *v_dev, int64_t n, double *max_host) {
    if (!ctx || !v_dev || !max_host) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    double *work = static_cast<double *>(ctx->work->p);
    HIP_TRY(launch_reduce_max(v_dev, n, work + 1025, work, ctx->stream));
    HIP_TRY(hipMemcpyAsync(max_host, work + 1025, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_vector_sum_dev(ppca_ctx *ctx, const double *v_dev, const double *w_dev, int64_t n, double *sum_host) {
    if (!ctx || !v_dev || !sum_host) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    double *work = static_cast<double *>(ctx->work->p);
    HIP_TRY(launch_reduce_sum(v_dev, w_dev, n, work + 1026, work, ctx->stream));
    HIP_TRY(hipMemcpyAsync(sum_host, work + 1026, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_vector_exp_shift_dev(ppca_ctx *ctx, const double *v_dev, double shift, int64_t n, double *out_dev) {
    if (!ctx || !v_dev || !out_dev) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    double *work = static_cast<double *>(ctx->work->p);
    HIP_TRY(hipMemcpyAsync(work + 1025, &shift, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(launch_exp_shift(v_dev, work + 1025, n, out_dev, ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_mix_reconstruct(ppca_ctx *ctx, ppca_dataset *ds, ppca_model *const *models, const double *log_weights,
                                    int32_t n_models, int32_t mode, ppca_dataset **out) {
    if (!ctx || !log_weights || !out) return fail(PPCA_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 3) return fail(PPCA_ERR_INVALID, "mode must be 0..3");
    if (int rc = mix_check(ds, models, n_models)) return rc;
    USE_CTX(ctx);
    const int64_t n = ds->n;
    const int d = ds->d, nm = n_models;
    auto nd = std::make_unique<ppca_dataset>();
    nd->ctx = ctx;
    nd->n = n;
    nd->d = d;  // weights are NOT carried over: mix.rs:245-265 collects fresh samples
    if (int rc = dev_alloc(sizeof(double) * (size_t)std::max<int64_t>(n, 1) * d, &nd->xbuf)) return rc;
    nd->X = static_cast<const double *>(nd->xbuf->p);
    if (n == 0) {
        *out = nd.release();
        return PPCA_OK;
    }
    BufRef llk, u, lse, lp;
    if (int rc = mix_posteriors(ctx, ds, models, log_weights, nm, llk, u, lse, &lp)) return rc;
    const double *logpost = static_cast<const double *>(lp->p);
    double *o = static_cast<double *>(nd->xbuf->p);
    struct Tmp {  // frees the per-component output on every exit path
        ppca_dataset *p = nullptr;
        ~Tmp() { if (p) ppca_dataset_free(p); }
    };
    const int vmode = mode & 1;  // 0 smooth-like, 1 extrapolate-like
    BufRef meanbuf;
    double *mean = nullptr;
    if (mode >= 2) {  // the diagonals are taken around the mixture mean (:448, :495): first pass
        if (int rc = dev_alloc(sizeof(double) * (size_t)n * d, &meanbuf)) return rc;
        mean = static_cast<double *>(meanbuf->p);
    }
    double *first_target = mode >= 2 ? mean : o;
    for (int c = 0; c < nm; ++c) {
        Tmp val;
        if (int rc = recon_common(ctx, ds, models[c], vmode, &val.p)) return rc;
        HIP_TRY(launch_mix_accumulate(first_target, val.p->X, nullptr, nullptr, logpost, c, nm, n, d, c == 0, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (mode >= 2) {
        for (int c = 0; c < nm; ++c) {
            Tmp val, dg;
            if (int rc = recon_common(ctx, ds, models[c], vmode, &val.p)) return rc;
            if (int rc = recon_common(ctx, ds, models[c], 2 + vmode, &dg.p)) return rc;
            HIP_TRY(launch_mix_accumulate(o, dg.p->X, val.p->X, mean, logpost, c, nm, n, d, c == 0, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    }
    *out = nd.release();
    return PPCA_OK;
}

// ------------------------------------------------------------------ debug
// Every launcher sizes its grid from ctx->n_cu (persistent workgroups, one per CU): capping it makes a workgroup walk many
// tiles at a size the CPU oracle still finishes in seconds.
extern "C" int ppca_ctx_set_grid_limit(ppca_ctx *ctx, int32_t n_workgroups) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "null context");
    USE_CTX(ctx);
    ctx->n_cu = (n_workgroups > 0 && n_workgroups < ctx->n_cu_device) ? n_workgroups : ctx->n_cu_device;
    return PPCA_OK;
}

extern "C" int ppca_ctx_set_heavy_rows(ppca_ctx *ctx, int32_t max_rows) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "null context");
    USE_CTX(ctx);
    ctx->heavy_max = max_rows < 0 ? 0 : (max_rows > 32 ? 32 : max_rows);
    return PPCA_OK;
}

extern "C" int ppca_em_last_guard(ppca_ctx *ctx, int32_t *gram_unsafe, int32_t *stats_unsafe) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "null context");
    USE_CTX(ctx);
    // (the verdicts reduce_wguard_kernel filed for the PASS: the tile flags themselves may already be the next model's --
    //  finalize_qprep_kernel builds its table at the end of the step.  After a multi-component mixture step: OR over its K passes.
    //  With a pinned engine -- PPCA_GRAM_FP64=1, tuning builds -- nothing files a verdict: both read 0.)
    int g = 0, w = 0;
    std::vector<int> flags(16 * std::max<size_t>(1, ctx->guard_words.size()), 0);
    for (size_t i = 0; i < ctx->guard_words.size(); ++i)
        HIP_TRY(hipMemcpyAsync(flags.data() + 16 * i, ctx->guard_words[i], sizeof(int) * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->guard_words.size(); ++i) {
        g |= flags[16 * i + QF_GVERDICT];
        w |= flags[16 * i + QF_WVERDICT];
    }
    if (gram_unsafe) *gram_unsafe = g;
    if (stats_unsafe) *stats_unsafe = w;
    return PPCA_OK;
}

extern "C" int ppca_em_last_fallback(ppca_ctx *ctx, int32_t *mode, int32_t *workgroups, int64_t *rows, double *stage_ms) {
    if (!ctx) return fail(PPCA_ERR_INVALID, "null context");
    USE_CTX(ctx);
    std::vector<int> flags(16 * std::max<size_t>(1, ctx->guard_words.size()), 0);
    for (size_t i = 0; i < ctx->guard_words.size(); ++i)
        HIP_TRY(hipMemcpyAsync(flags.data() + 16 * i, ctx->guard_words[i], sizeof(int) * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int md = 0, wg = 0;
    int64_t rw = 0;
    for (size_t i = 0; i < ctx->guard_words.size(); ++i) {  // (a multi-component mixture step: the largest mode, the sums over its K passes)
        md = std::max(md, flags[16 * i + QF_MODE]);
        wg += flags[16 * i + QF_NFLAGGED];
        rw += flags[16 * i + QF_NROWS2];
    }
    if (mode) *mode = md;
    if (workgroups) *workgroups = wg;
    if (rows) *rows = rw;
    double tot = 0.0;
    for (auto &ev : ctx->events2) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    ctx->events2.clear();
    if (stage_ms) *stage_ms = tot;
    return PPCA_OK;
}

extern "C" int ppca_debug_counters(ppca_ctx *ctx, int64_t *out8, int32_t reset) {
    if (!ctx || !out8) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(em8_debug_counters(c, reset, ctx->stream));
    {  // (PPCA_EM9=1: the pipelined-solve variant keeps its own; reported in the same slots)
        unsigned long long c9[4] = {0, 0, 0, 0};
        HIP_TRY(em9_debug_counters(c9, reset, ctx->stream));
        c[0] += c9[0];
        c[1] += c9[1];
        c[2] = std::max(c[2], c9[2]);
        c[3] += c9[3];
    }
    HIP_TRY(em16_debug_counters(c + 4, reset, ctx->stream));
    for (int i = 0; i < 8; ++i) out8[i] = (int64_t)c[i];
    return PPCA_OK;
}

extern "C" int ppca_gram_engine(ppca_ctx *ctx, const ppca_model *model, int32_t *engine) {
    if (!ctx || !model || !engine) return fail(PPCA_ERR_INVALID, "null argument");
    if (int rc = check_path(model->d, model->k)) return rc;
    USE_CTX(ctx);
    if (ppca_path_kind(model->d, model->k) == 0) {
        // the generic pipeline: int8-sliced contractions behind the same per-model guard, fp64 MFMA when it trips (or
        // under PPCA_GENERIC_FP64=1)
        if (int rc = ensure(ctx->gws, ctx->gws_cap, generic_workspace_bytes(model->d, model->k, 1))) return rc;
        const int *flag_dev = nullptr;
        int forced = -1;
        HIP_TRY(generic_gram_guard(model->d, model->k, model->p(), ctx->gws->p, ctx->stream, &flag_dev, &forced));
        if (forced >= 0) {
            *engine = forced;
            return PPCA_OK;
        }
        int flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, flag_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        *engine = flag ? 1 : 0;
        return PPCA_OK;
    }
    PassArgs a{};
    a.model = model->p();
    a.d = model->d;
    if (int rc = qtab_bind(ctx, model, a)) return rc;  // (launch_gram_guard always builds the table: it is this model's afterwards)
    int forced = -1;
    HIP_TRY(launch_gram_guard(model->k, a, ctx->stream, &forced));
    qtab_commit(ctx, model, forced < 0);  // (a pinned engine builds no table)
    if (forced >= 0) {
        *engine = forced;
        return PPCA_OK;
    }
    int flags[8];
    HIP_TRY(hipMemcpyAsync(flags, a.qflag, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int unsafe = 0;
    for (int t = 0; t < fused_gram_tiles(model->k); ++t) unsafe |= flags[t];
    *engine = unsafe ? 1 : 0;
    return PPCA_OK;
}

extern "C" int ppca_debug_mfma_probe(ppca_ctx *ctx, const double *a16x4, const double *b4x16, double *out16x16) {
    if (!ctx || !a16x4 || !b4x16 || !out16x16) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    BufRef buf;
    if (int rc = dev_alloc(sizeof(double) * (64 + 64 + 256), &buf)) return rc;
    double *p = static_cast<double *>(buf->p);
    HIP_TRY(hipMemcpyAsync(p, a16x4, sizeof(double) * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(p + 64, b4x16, sizeof(double) * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_mfma_probe(p, p + 64, p + 128, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out16x16, p + 128, sizeof(double) * 256, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}

extern "C" int ppca_debug_mfma_i8_probe(ppca_ctx *ctx, const int8_t *a_regs, const int8_t *b_regs, int32_t *out_regs) {
    if (!ctx || !a_regs || !b_regs || !out_regs) return fail(PPCA_ERR_INVALID, "null argument");
    USE_CTX(ctx);
    BufRef buf;
    if (int rc = dev_alloc(1024 + 1024 + 1024, &buf)) return rc;
    char *p = static_cast<char *>(buf->p);
    HIP_TRY(hipMemcpyAsync(p, a_regs, 1024, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(p + 1024, b_regs, 1024, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_mfma_i8_probe(reinterpret_cast<int *>(p), reinterpret_cast<int *>(p + 1024),
                                 reinterpret_cast<int *>(p + 2048), ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_regs, p + 2048, 1024, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PPCA_OK;
}
