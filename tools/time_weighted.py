"""EM pass of the fused shapes on an unweighted dataset, a weighted one (em8_kernel<K, false, true>) and through the
mixture's gathered form is not reachable from here: times ppca_em_accumulate per call (diagnostic)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib

n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.default_context()
truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
spec = _lib.SynthSpec(0, n, d, k, 0.1, 0.3, 0, 0, 1033, truth._c.ctypes.data_as(_lib.c_double_p), truth._mean.ctypes.data_as(_lib.c_double_p))
h = C.c_void_p()
_lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
ds = P.Dataset._wrap(h, ctx)
dsw = ds.with_weights(np.random.default_rng(3).uniform(0.5, 1.5, n))
m = P.PPCAModel.init(k, ds, seed=3).iterate(ds)
for name, dd in (("unweighted", ds), ("weighted", dsw)):
    mm = m
    for _ in range(2): mm = mm.iterate(dd)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5): mm = mm.iterate(dd)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:12s} {dt*1e3:8.3f} ms per EM iteration  (N = {n})", flush=True)
