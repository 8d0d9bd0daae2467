"""Soak of the split pipeline at config 4's shape (N = 400 k): forty statistics passes must be bit-identical (a race in the int8 GEMM's ring would show),
then ten EM iterations.  Diagnostic; run on the GPU box:  python tools/soak_generic.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
n, d, k = 400_000, 1024, 64
truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
ds = truth.sample(n, 0.5, seed=3)
m = P.PPCAModel.init(k, ds, seed=4)
L = _lib.lib().ppca_stats_len(d, k)
ref = None
t = time.time()
for it in range(40):
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    if ref is None: ref = got.copy()
    assert np.array_equal(ref, got), ("not bit-reproducible at repeat", it)
print("40 statistics passes at N = 400 k, d = 1024, k = 64: bit-identical;", round(time.time() - t, 1), "s; guard", ds._ctx.last_guard())
for it in range(10):
    m = m.iterate(ds)
print("10 EM iterations: sigma", m.isotropic_noise)
