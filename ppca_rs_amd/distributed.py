"""Sample-sharded EM across GPUs: one process per GPU, ONE collective per iteration.

The dataset is split into contiguous row blocks of ceil(N / world) rows (the rule
of Dataset.chunks, src/python_bindings.rs:110-118).  Every M-step statistic is a
weighted sum over samples (ppca_model.rs:281-358), so an EM iteration is
    local fused pass  ->  all-reduce(sum) of the packed statistics  ->  identical
    finalisation on every rank (no broadcast needed).
`torch.distributed` supplies the collective: backend "nccl" (= RCCL over xGMI) for
device buffers, "gloo" for host buffers (CPU tests).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import check, lib, ptr
from .api import Dataset, PPCAModel, Prior, _DevModel, _prior_ref


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of rank's contiguous block; blocks are ceil(n / world) rows."""
    stride = int(math.ceil(n / world)) if world > 0 else n
    start = min(n, rank * stride)
    return start, min(n, start + stride)


def stats_len(d: int, k: int) -> int:
    return int(lib().ppca_stats_len(d, k))


def finalize_host(model: PPCAModel, stats: np.ndarray, prior: Optional[Prior] = None) -> PPCAModel:
    """M-step finalisation from (already reduced) host statistics -- pure host code."""
    d, k = model.output_size, model.state_size
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    if stats.shape[0] != stats_len(d, k):
        raise ValueError("statistics buffer has the wrong length")
    c_out = np.empty((d, k))
    m_out = np.empty(d)
    sig = C.c_double(0.0)
    pref, keep = _prior_ref(prior)
    check(lib().ppca_em_finalize_host(d, k, model.isotropic_noise, ptr(model._c), ptr(model._mean), ptr(stats), pref,
                                      C.byref(sig), ptr(c_out), ptr(m_out)))
    return PPCAModel(sig.value, c_out, m_out)


def allreduce_finalize_host(model: PPCAModel, local_stats: np.ndarray, prior: Optional[Prior] = None, group=None):
    """Host flavour of one distributed M-step: all-reduce the shard statistics over
    the default process group (gloo) and finalise.  Returns (model', llk of model)."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(local_stats, dtype=np.float64).copy())
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    red = t.numpy()
    d, k = model.output_size, model.state_size
    llk = float(red[stats_len(d, k) - 8 + 2])
    return finalize_host(model, red, prior), llk


class Communicator:
    """ppca_comm: the library's own RCCL communicator over the ranks (one rank per GPU), bound to a context.

    `Communicator.from_torch(ctx)` takes rank / world size from the initialised torch.distributed process group and
    ships rank 0's 128-byte unique id through it (any backend: the id travels as a pickled object); the data path
    -- the all-reduce of the statistics buffer -- then runs inside the library on the context stream."""

    def __init__(self, ctx: _lib.Context, n_ranks: int, rank: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        self.ctx, self.n_ranks, self.rank = ctx, n_ranks, rank
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        check(lib().ppca_comm_create(ctx.handle, n_ranks, rank, buf, C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * 128)()
        check(lib().ppca_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def backend() -> str:
        return lib().ppca_comm_backend().decode()

    @classmethod
    def from_torch(cls, ctx: _lib.Context, group=None) -> "Communicator":
        """Collective over the process group, failures included: rank 0's unique id -- or the reason it could not make
        one (librccl missing, bound to another HIP runtime) -- is what gets broadcast, and after ncclCommInitRank every
        rank learns whether ALL ranks succeeded; on any failure every rank raises the same error after leaving the
        collectives, so a caller's fallback (bench.py --collective auto) is taken by all of them together."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return cls(ctx, 1, 0, cls.unique_id())
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            try:
                box = [("id", cls.unique_id())]
            except Exception as e:  # noqa: BLE001
                box = [("error", f"rank 0: {e}")]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        kind, payload = box[0]
        comm, err = None, None
        if kind == "id":
            try:
                comm = cls(ctx, world, rank, payload)
            except Exception as e:  # noqa: BLE001
                err = f"rank {rank}: {e}"
        else:
            err = payload
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)
        errs = [e for e in errs if e]
        if errs:
            if comm is not None:
                comm.close()
            raise RuntimeError("library communicator unavailable: " + "; ".join(errs))
        return comm

    def allreduce(self, ptr_dev: int, n: int, op: str = "sum") -> None:
        check(lib().ppca_comm_allreduce(self.h, C.c_void_p(ptr_dev), n, {"sum": 0, "max": 1}[op]))

    def close(self) -> None:
        if self.h:
            lib().ppca_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedEM:
    """Device flavour: holds this rank's shard and a ping-pong pair of device models;
    `step()` enqueues pass -> all-reduce -> finalise on ONE dedicated stream, without any host
    synchronisation.  With `comm` (a Communicator) the whole step is ONE C-ABI call,
    ppca_em_step_sharded, and the collective is the library's own RCCL all-reduce; without it the
    collective is torch.distributed's (backend "nccl" = RCCL for device buffers, "gloo" stages through
    the host) on a dedicated torch stream that the library is pointed at.  (A dedicated
    stream, not torch's default one: the legacy default stream has handle 0, which the C-ABI
    reads as "create your own stream" -- the collective would then not be ordered after the
    kernels.)"""

    def __init__(self, shard: Dataset, start: PPCAModel, prior: Optional[Prior] = None, group=None, comm=None):
        import torch

        self.torch = torch
        self.shard, self.prior, self.group, self.comm = shard, prior, group, comm
        self.ctx = shard._ctx
        self.d, self.k = start.output_size, start.state_size
        self.stream = torch.cuda.Stream()
        assert self.stream.cuda_stream != 0
        with torch.cuda.stream(self.stream):
            self.stats = torch.zeros(stats_len(self.d, self.k), dtype=torch.float64, device="cuda")
        # Two PRIVATE device models: the ping-pong must never write into the cached device copy of the caller's
        # (immutable) start model.
        h = C.c_void_p()
        check(lib().ppca_model_create(self.ctx.handle, self.d, self.k, start.isotropic_noise, ptr(start._c),
                                      ptr(start._mean), C.byref(h)))
        self.cur = _DevModel(h)
        h = C.c_void_p()
        check(lib().ppca_model_alloc(self.ctx.handle, self.d, self.k, C.byref(h)))
        self.nxt = _DevModel(h)
        self._pref, self._keep = _prior_ref(prior)
        torch.cuda.synchronize()
        self.ctx.set_stream(self.stream.cuda_stream)

    def step(self) -> None:
        import torch.distributed as dist

        torch = self.torch
        if self.comm is not None:
            # statistics in the context's scratch; llk_of_previous() re-reads them from there
            check(lib().ppca_em_step_sharded(self.comm.h, self.shard._h, self.cur.h, self._pref, self.nxt.h, None))
            self.cur, self.nxt = self.nxt, self.cur
            return
        with torch.cuda.stream(self.stream):
            check(lib().ppca_em_accumulate(self.ctx.handle, self.shard._h, self.cur.h, C.c_void_p(self.stats.data_ptr())))
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                dist.all_reduce(self.stats, op=dist.ReduceOp.SUM, group=self.group)
            check(lib().ppca_em_finalize(self.ctx.handle, self.cur.h, C.c_void_p(self.stats.data_ptr()), self._pref,
                                         self.nxt.h))
        self.cur, self.nxt = self.nxt, self.cur

    def synchronize(self) -> None:
        self.stream.synchronize()

    def llk_of_previous(self) -> float:
        """llk of the model that entered the last step() (synchronises)."""
        if self.comm is not None:
            v = C.c_double(0.0)
            check(lib().ppca_em_last_llk(self.ctx.handle, C.byref(v)))
            return v.value
        self.stream.synchronize()
        return float(self.stats[stats_len(self.d, self.k) - 8 + 2].item())

    def model(self) -> PPCAModel:
        self.stream.synchronize()
        sig = C.c_double(0.0)
        c = np.empty((self.d, self.k))
        m = np.empty(self.d)
        check(lib().ppca_model_download(self.cur.h, C.byref(sig), ptr(c), ptr(m)))
        return PPCAModel(sig.value, c, m)

    def close(self) -> None:
        """Detach the library from the torch stream (before the stream object dies)."""
        self.stream.synchronize()
        self.ctx.set_stream(None)


# --------------------------------------------------------------------------- sharded mixture (BASELINE config 5)
class _DeviceMixBackend:
    """The per-shard pieces of one mixture EM step on this rank's GPU, through the C-ABI building blocks -- the form
    used when the collective is torch.distributed's (no library communicator).  Components may differ in state size
    (mix.rs:50-71): every component has its own statistics length and offset in the packed buffer."""

    def __init__(self, shard: Dataset, models, prior):
        import torch

        self.torch, self.shard, self.ctx, self.prior = torch, shard, shard._ctx, prior
        self.d, self.nm = models[0].output_size, len(models)
        self.ks = [m.state_size for m in models]
        self.cur = [m._device(self.ctx) for m in models]
        self._keep_models = list(models)
        self.n = len(shard)
        self.rows_used = [0] * len(models)  # rows each component pass gathered in the last step
        self.Ls = [stats_len(self.d, k) for k in self.ks]
        self.offs = np.concatenate([[0], np.cumsum(self.Ls)]).astype(np.int64)
        self.u = torch.empty(max(self.nm * self.n, 1), dtype=torch.float64, device="cuda")
        self.lse = torch.empty(max(self.n, 1), dtype=torch.float64, device="cuda")
        self.stats = torch.zeros(int(self.offs[-1]), dtype=torch.float64, device="cuda")
        self._pref, self._keep = _prior_ref(prior)

    def _arr(self):
        return (C.c_void_p * self.nm)(*[m.h for m in self.cur])

    def responsibilities(self, log_weights: np.ndarray) -> float:
        """u[c][i] = ln w_i + log r_ic on the device; returns this shard's share of the mixture llk."""
        if self.n == 0:
            return 0.0
        check(lib().ppca_mix_responsibilities_dev(self.ctx.handle, self.shard._h, self._arr(), ptr(log_weights), self.nm,
                                                  C.c_void_p(self.u.data_ptr()), C.c_void_p(self.lse.data_ptr())))
        tot = C.c_double(0.0)
        wdev = lib().ppca_dataset_device_weights(self.shard._h)
        check(lib().ppca_vector_sum_dev(self.ctx.handle, C.c_void_p(self.lse.data_ptr()), C.c_void_p(wdev) if wdev else None,
                                        self.n, C.byref(tot)))
        return tot.value

    def local_max(self, c: int) -> float:
        if self.n == 0:
            return -np.inf
        mx = C.c_double(0.0)
        check(lib().ppca_vector_max_dev(self.ctx.handle, C.c_void_p(self.u.data_ptr() + 8 * c * self.n), self.n, C.byref(mx)))
        return mx.value

    def accumulate(self, c: int, shift: float):
        """weights exp(u_c - shift), their sum, and the component's weighted statistics (device tensor view); rows whose
        weight is below 2^-200 of the component's largest are dropped from the pass (ppca_mix_component_stats)."""
        view = self.stats[int(self.offs[c]):int(self.offs[c + 1])]
        if self.n == 0:
            view.zero_()
            self.rows_used[c] = 0
            return 0.0, view
        s, used = C.c_double(0.0), C.c_int64(0)
        check(lib().ppca_mix_component_stats(self.ctx.handle, self.shard._h, self.cur[c].h,
                                             C.c_void_p(self.u.data_ptr() + 8 * c * self.n), shift,
                                             C.c_void_p(view.data_ptr()), C.byref(s), C.byref(used)))
        self.rows_used[c] = used.value  # (what ppca_mix_last_rows_used reports for the one-call path)
        return s.value, view

    def pack(self, extras: np.ndarray):
        return self.torch.cat([self.stats, self.torch.from_numpy(extras).to("cuda")])

    def unpack_extras(self, packed, count: int) -> np.ndarray:
        return packed[-count:].cpu().numpy()

    def finalize(self, c: int, packed) -> None:
        h = C.c_void_p()
        check(lib().ppca_model_alloc(self.ctx.handle, self.d, self.ks[c], C.byref(h)))
        new = _DevModel(h)
        check(lib().ppca_em_finalize(self.ctx.handle, self.cur[c].h, C.c_void_p(packed.data_ptr() + 8 * int(self.offs[c])),
                                     self._pref, new.h))
        self.cur[c] = new

    def max_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device="cuda")

    def models(self):
        return [PPCAModel._from_device(m, self.ctx, self.d, k) for m, k in zip(self.cur, self.ks)]


class ShardedMixEM:
    """PPCAMix::iterate_with_prior (mix.rs:281-337) over row shards, one process per GPU.

    With `comm` (a Communicator) -- or on a single rank -- a step is ONE C-ABI call, ppca_mix_em_step_sharded /
    ppca_mix_em_step: local responsibilities, all-reduce(MAX) of the K per-component maxima of ln w_i + log r_ic
    (:312-317 take the maximum over ALL samples), the K gathered component passes, ONE all-reduce(SUM) of
    [K statistic buffers | K weight sums | llk], identical finalisation and new log-weights (:335) on every rank --
    enqueued on the context stream with the shifts computed on the device.

    Without a communicator on several ranks (the collective is torch.distributed's: gloo in the CPU tests, or
    `--collective torch`), or with an injected `backend` (the CPU stand-in of the tests), the same protocol is
    orchestrated here from the C-ABI building blocks."""

    def __init__(self, shard, start, prior: Optional[Prior] = None, group=None, backend=None, comm=None):
        self.group = group
        self.comm = comm
        self.log_weights = np.array(start.log_weights, dtype=np.float64)
        self.nm = len(start.models)
        self.backend = backend
        self._one_call = backend is None and (comm is not None or not self._multi())
        if self._one_call:
            self.shard, self.ctx = shard, shard._ctx
            self.d, self.ks = start.models[0].output_size, [m.state_size for m in start.models]
            self._pref, self._keep = _prior_ref(prior)
            self.cur, self.nxt = [], []
            for m in start.models:  # two PRIVATE device models per component (ping-pong; the caller's stay untouched)
                h = C.c_void_p()
                check(lib().ppca_model_create(self.ctx.handle, self.d, m.state_size, m.isotropic_noise, ptr(m._c), ptr(m._mean),
                                              C.byref(h)))
                self.cur.append(_DevModel(h))
                h = C.c_void_p()
                check(lib().ppca_model_alloc(self.ctx.handle, self.d, m.state_size, C.byref(h)))
                self.nxt.append(_DevModel(h))
        elif backend is None:
            self.backend = _DeviceMixBackend(shard, start.models, prior)

    def _multi(self) -> bool:
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def step(self) -> float:
        """One EM step; returns the mixture log-likelihood of the input mixture over ALL shards."""
        if self._one_call:
            arr_in = (C.c_void_p * self.nm)(*[m.h for m in self.cur])
            arr_out = (C.c_void_p * self.nm)(*[m.h for m in self.nxt])
            lw_out = np.empty(self.nm)
            llk = C.c_double(0.0)
            if self.comm is not None:
                check(lib().ppca_mix_em_step_sharded(self.comm.h, self.shard._h, arr_in, ptr(self.log_weights), self.nm, self._pref,
                                                     arr_out, ptr(lw_out), C.byref(llk)))
            else:
                check(lib().ppca_mix_em_step(self.ctx.handle, self.shard._h, arr_in, ptr(self.log_weights), self.nm, self._pref,
                                             arr_out, ptr(lw_out), C.byref(llk)))
            self.cur, self.nxt = self.nxt, self.cur
            self.log_weights = lw_out
            return llk.value
        import torch.distributed as dist

        be, nm = self.backend, self.nm
        multi = self._multi()
        llk_local = be.responsibilities(self.log_weights)
        mx = be.max_tensor([be.local_max(c) for c in range(nm)])
        if multi:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        mx = np.asarray(mx.cpu().numpy(), dtype=np.float64)
        shifts = np.where(np.isfinite(mx), mx, 0.0)
        sums = np.zeros(nm)
        for c in range(nm):
            sums[c], _ = be.accumulate(c, float(shifts[c]))
        packed = be.pack(np.concatenate([sums, [llk_local]]))
        if multi:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        extras = be.unpack_extras(packed, nm + 1)
        for c in range(nm):
            be.finalize(c, packed)
        with np.errstate(divide="ignore"):
            logsum = np.log(extras[:nm]) + shifts
        m = logsum.max()
        self.log_weights = logsum - m - np.log(np.exp(logsum - m).sum())  # robust_log_softmax, mix.rs:14-18
        return float(extras[nm])

    def mixture(self):
        from .api import PPCAMix

        if self._one_call:
            # host copies only: cur / nxt are this object's ping-pong buffers and are overwritten two steps later, so a
            # model handed out mid-training must never carry one of them as its cached device handle
            models = []
            for dev, k in zip(self.cur, self.ks):
                sig, c, mean = C.c_double(0.0), np.empty((self.d, k)), np.empty(self.d)
                check(lib().ppca_model_download(dev.h, C.byref(sig), ptr(c), ptr(mean)))
                models.append(PPCAModel(sig.value, c, mean))
            return PPCAMix(models, np.array(self.log_weights, dtype=np.float64))
        return PPCAMix(self.backend.models(), self.log_weights)
