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


class ShardedEM:
    """Device flavour: holds this rank's shard and a ping-pong pair of device models;
    `step()` enqueues pass -> all-reduce -> finalise on ONE dedicated torch stream, which is
    also the stream the library launches on, without any host synchronisation.  (A dedicated
    stream, not torch's default one: the legacy default stream has handle 0, which the C-ABI
    reads as "create your own stream" -- the collective would then not be ordered after the
    kernels.)"""

    def __init__(self, shard: Dataset, start: PPCAModel, prior: Optional[Prior] = None, group=None):
        import torch

        self.torch = torch
        self.shard, self.prior, self.group = shard, prior, group
        self.ctx = shard._ctx
        self.d, self.k = start.output_size, start.state_size
        self.stream = torch.cuda.Stream()
        assert self.stream.cuda_stream != 0
        with torch.cuda.stream(self.stream):
            self.stats = torch.zeros(stats_len(self.d, self.k), dtype=torch.float64, device="cuda")
        self.cur = start._device(self.ctx)
        h = C.c_void_p()
        check(lib().ppca_model_alloc(self.ctx.handle, self.d, self.k, C.byref(h)))
        self.nxt = _DevModel(h)
        self._pref, self._keep = _prior_ref(prior)
        self._start = start  # keeps the first device model alive
        torch.cuda.synchronize()
        self.ctx.set_stream(self.stream.cuda_stream)

    def step(self) -> None:
        import torch.distributed as dist

        torch = self.torch
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
