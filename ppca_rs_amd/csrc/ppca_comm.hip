// ppca_comm.hip -- the ONE collective of the path behind the C-ABI: a sum of the packed sufficient statistics over
// the sample shards (replaces the in-process rayon reductions of ppca/src/ppca_model.rs:290-293, :350-358), on RCCL
// over xGMI, enqueued on the context stream between the shard's pass and the (replicated) finalisation.
//
// librccl is resolved at run time (dlopen) so that the library loads, and everything single-GPU works, on a host
// without it; the first ppca_comm_* call fails with PPCA_ERR_UNSUPPORTED and a message naming what was tried.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the table below

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ppca_handles.hpp"

using namespace ppca;
using namespace ppca_host;

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string origin, error;
    bool ok = false;
};

// A library that is already in the process (torch ships and loads its own librccl) is reused; otherwise the
// loader's search path, then the ROCm default.  PPCA_RCCL_LIB overrides.  A candidate is accepted only if it runs
// on the SAME HIP runtime as this library: a PyTorch wheel bundles its own libamdhip64 next to its librccl, and
// depending on the import order this library is bound either to that one or to /opt/rocm's -- streams and events
// of one runtime mean nothing to the other.
const Rccl &rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::pair<std::string, int>> tries;
        if (const char *e = getenv("PPCA_RCCL_LIB")) tries.push_back({e, RTLD_NOW | RTLD_GLOBAL});
        for (const char *n : {"librccl.so.1", "librccl.so"}) tries.push_back({n, RTLD_NOW | RTLD_NOLOAD});
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            tries.push_back({n, RTLD_NOW | RTLD_GLOBAL});
        void *h = nullptr;
        void *const my_hip = reinterpret_cast<void *>(&hipStreamSynchronize);
        for (auto &t : tries) {
            h = dlopen(t.first.c_str(), t.second);
            if (h) {
                // dlsym on a handle searches the library's dependency tree: the HIP runtime THIS librccl calls into
                void *its_hip = dlsym(h, "hipStreamSynchronize");
                if (its_hip == my_hip) {
                    Dl_info info;
                    void *probe = dlsym(h, "ncclAllReduce");
                    api.origin = (probe && dladdr(probe, &info) && info.dli_fname) ? info.dli_fname : t.first;
                    break;
                }
                api.error += t.first + " (bound to another HIP runtime); ";
                h = nullptr;
                continue;
            }
            api.error += t.first + "; ";
        }
        if (!h) {
            api.error = "librccl not found (tried: " + api.error + ")";
            return;
        }
        bool all = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(h, name);
            if (!p) {
                all = false;
                api.error += std::string("missing symbol ") + name + "; ";
            }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        api.ok = all;
    });
    return api;
}

int need_rccl(const Rccl **out) {
    const Rccl &r = rccl();
    if (!r.ok) return fail(PPCA_ERR_UNSUPPORTED, "RCCL is not available: %s", r.error.c_str());
    *out = &r;
    return PPCA_OK;
}

#define RCCL_TRY(r, expr)                                                                                      \
    do {                                                                                                       \
        ncclResult_t _n = (expr);                                                                              \
        if (_n != ncclSuccess) return fail(PPCA_ERR_HIP, "%s failed: %s", #expr, (r)->GetErrorString(_n));     \
    } while (0)

}  // namespace

struct ppca_comm {
    ppca_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
};

static_assert(PPCA_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int ppca_comm_unique_id(void *id_out) {
    if (!id_out) return fail(PPCA_ERR_INVALID, "null argument");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    ncclUniqueId id;
    RCCL_TRY(r, r->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return PPCA_OK;
}

extern "C" int ppca_comm_create(ppca_ctx *ctx, int32_t n_ranks, int32_t rank, const void *unique_id, ppca_comm **out) {
    if (!ctx || !unique_id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(PPCA_ERR_INVALID, "bad communicator arguments");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    USE_CTX(ctx);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    auto c = std::make_unique<ppca_comm>();
    c->ctx = ctx;
    c->n_ranks = n_ranks;
    c->rank = rank;
    RCCL_TRY(r, r->CommInitRank(&c->comm, n_ranks, id, rank));
    *out = c.release();
    return PPCA_OK;
}

extern "C" int ppca_comm_create_all(ppca_ctx *const *ctxs, int32_t n, ppca_comm **out) {
    if (!ctxs || !out || n < 1) return fail(PPCA_ERR_INVALID, "bad communicator arguments");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(PPCA_ERR_INVALID, "null context");
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return fail(PPCA_ERR_INVALID, "contexts %d and %d share device %d", j, i, devs[i]);
    }
    std::vector<ncclComm_t> comms(n);
    RCCL_TRY(r, r->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        auto *c = new ppca_comm();
        c->ctx = ctxs[i];
        c->comm = comms[i];
        c->n_ranks = n;
        c->rank = i;
        out[i] = c;
    }
    return PPCA_OK;
}

extern "C" int ppca_comm_destroy(ppca_comm *comm) {
    if (!comm) return PPCA_OK;
    const Rccl &r = rccl();
    if (r.ok && comm->comm) {
        (void)hipSetDevice(comm->ctx->device);
        (void)hipStreamSynchronize(comm->ctx->stream);
        (void)r.CommDestroy(comm->comm);
    }
    delete comm;
    return PPCA_OK;
}

ppca_ctx *ppca_comm_context(ppca_comm *comm) { return comm ? comm->ctx : nullptr; }

extern "C" int32_t ppca_comm_n_ranks(const ppca_comm *comm) { return comm ? comm->n_ranks : 0; }
extern "C" int32_t ppca_comm_rank(const ppca_comm *comm) { return comm ? comm->rank : -1; }

extern "C" const char *ppca_comm_backend(void) {
    static thread_local std::string s;
    const Rccl &r = rccl();
    if (!r.ok) {
        s = "unavailable: " + r.error;
    } else {
        int v = 0;
        (void)r.GetVersion(&v);
        s = "rccl " + std::to_string(v) + " via " + r.origin;
    }
    return s.c_str();
}

extern "C" int ppca_comm_allreduce(ppca_comm *comm, double *buf_dev, int64_t n, int32_t op) {
    if (!comm || (!buf_dev && n > 0) || n < 0 || (op != 0 && op != 1)) return fail(PPCA_ERR_INVALID, "bad all-reduce arguments");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    USE_CTX(comm->ctx);
    if (n == 0) return PPCA_OK;
    RCCL_TRY(r, r->AllReduce(buf_dev, buf_dev, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, comm->comm,
                             comm->ctx->stream));
    return PPCA_OK;
}

// pass over the shard -> all-reduce(sum) of the packed statistics -> finalisation, all on the context stream
extern "C" int ppca_em_step_sharded(ppca_comm *comm, ppca_dataset *shard, const ppca_model *model_in,
                                    const ppca_prior *prior, ppca_model *out, double *llk_in) {
    if (!comm || !shard || !model_in || !out) return fail(PPCA_ERR_INVALID, "null argument");
    ppca_ctx *ctx = comm->ctx;
    USE_CTX(ctx);
    const StatsLayout L(model_in->d, model_in->k);
    if (int rc = ensure(ctx->stats, ctx->stats_cap, sizeof(double) * (size_t)L.len)) return rc;
    double *stats = static_cast<double *>(ctx->stats->p);
    if (int rc = ppca_em_accumulate(ctx, shard, model_in, stats)) return rc;  // an empty shard contributes zeros
    if (int rc = ppca_comm_allreduce(comm, stats, L.len, 0)) return rc;
    if (int rc = ppca_host::em_finalize_with_table(ctx, model_in, stats, prior, out)) return rc;
    ctx->stats_llk_at = L.scalars + SC_LLK;
    if (llk_in) {
        HIP_TRY(hipMemcpyAsync(llk_in, stats + L.scalars + SC_LLK, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PPCA_OK;
}

// The same for ONE host thread driving n devices (communicators of ppca_comm_create_all): the n all-reduces are
// issued inside one RCCL group so that no rank waits for a peer that the same thread has not enqueued yet.
extern "C" int ppca_em_step_group(ppca_comm *const *comms, int32_t n, ppca_dataset *const *shards,
                                  ppca_model *const *models_in, const ppca_prior *prior, ppca_model *const *models_out,
                                  double *llk_in) {
    if (!comms || !shards || !models_in || !models_out || n < 1) return fail(PPCA_ERR_INVALID, "null argument");
    const Rccl *r = nullptr;
    if (int rc = need_rccl(&r)) return rc;
    for (int i = 0; i < n; ++i)
        if (!comms[i] || !shards[i] || !models_in[i] || !models_out[i]) return fail(PPCA_ERR_INVALID, "null argument");
    const StatsLayout L(models_in[0]->d, models_in[0]->k);
    // Everything that can be refused is refused HERE, before any lock is taken or anything is enqueued on any device:
    // a failure half-way through the launch loop would leave the earlier devices with a pass in flight and an
    // all-reduce their peers never join.
    for (int i = 0; i < n; ++i) {
        if (comms[i]->n_ranks != n) return fail(PPCA_ERR_INVALID, "communicator %d belongs to a clique of %d, not %d", i, comms[i]->n_ranks, n);
        if (models_in[i]->d != L.d || models_in[i]->k != L.k || models_out[i]->d != L.d || models_out[i]->k != L.k)
            return fail(PPCA_ERR_INVALID, "model shapes differ (rank %d)", i);
        if (models_out[i] == models_in[i] || models_out[i]->buf == models_in[i]->buf)
            return fail(PPCA_ERR_INVALID, "out may not alias model_in (rank %d)", i);
        if (shards[i]->d != L.d) return fail(PPCA_ERR_INVALID, "shard %d has %d dimensions but the model has output size %d", i, shards[i]->d, L.d);
        const int dev = comms[i]->ctx->device;
        if (shards[i]->ctx->device != dev || models_in[i]->ctx->device != dev || models_out[i]->ctx->device != dev)
            return fail(PPCA_ERR_INVALID, "shard / models of rank %d do not live on its communicator's device %d", i, dev);
        for (int j = 0; j < i; ++j)
            if (comms[j]->ctx->device == dev) return fail(PPCA_ERR_INVALID, "communicators %d and %d share device %d", j, i, dev);
    }
    if (ppca_path_kind(L.d, L.k) < 0) return fail(PPCA_ERR_UNSUPPORTED, "state size %d is not supported", L.k);
    std::vector<std::unique_lock<std::recursive_mutex>> locks;
    for (int i = 0; i < n; ++i) locks.emplace_back(comms[i]->ctx->mu);
    for (int i = 0; i < n; ++i) {
        ppca_ctx *ctx = comms[i]->ctx;
        if (int rc = use_device(ctx)) return rc;
        if (int rc = ensure(ctx->stats, ctx->stats_cap, sizeof(double) * (size_t)L.len)) return rc;
        if (int rc = ppca_em_accumulate(ctx, shards[i], models_in[i], static_cast<double *>(ctx->stats->p))) return rc;
    }
    RCCL_TRY(r, r->GroupStart());
    for (int i = 0; i < n; ++i) {
        ppca_ctx *ctx = comms[i]->ctx;
        double *stats = static_cast<double *>(ctx->stats->p);
        ncclResult_t e = r->AllReduce(stats, stats, (size_t)L.len, ncclDouble, ncclSum, comms[i]->comm, ctx->stream);
        if (e != ncclSuccess) {
            (void)r->GroupEnd();
            return fail(PPCA_ERR_HIP, "ncclAllReduce failed: %s", r->GetErrorString(e));
        }
    }
    RCCL_TRY(r, r->GroupEnd());
    for (int i = 0; i < n; ++i) {
        ppca_ctx *ctx = comms[i]->ctx;
        if (int rc = ppca_host::em_finalize_with_table(ctx, models_in[i], static_cast<double *>(ctx->stats->p), prior, models_out[i])) return rc;
        ctx->stats_llk_at = L.scalars + SC_LLK;
    }
    if (llk_in) {
        ppca_ctx *ctx = comms[0]->ctx;
        if (int rc = use_device(ctx)) return rc;
        HIP_TRY(hipMemcpyAsync(llk_in, static_cast<double *>(ctx->stats->p) + L.scalars + SC_LLK, sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return PPCA_OK;
}
