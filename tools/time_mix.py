"""Times PPCAMix EM iterations (BASELINE config 5: K components, N x d, k) through the public API (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib

n, d, k, nm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = _lib.default_context()
rng = np.random.default_rng(1)
parts = []
for c in range(nm):
    truth = P.PPCAModel(0.1, rng.standard_normal((d, k)), 3.0 * rng.standard_normal(d))
    parts.append(truth.sample(n // nm, 0.3, seed=100 + c))
ds = P.Dataset.concat(parts)
del parts
mix = P.PPCAMix.init(nm, k, ds, seed=7)
prev = -np.inf
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 4
for it in range(iters):
    ctx.synchronize(); t0 = time.perf_counter()
    mix, llk = mix.iterate_with_llk(ds)
    ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"iter {it}: {dt*1e3:.1f} ms, llk/N of input mixture {llk/len(ds):.4f}", flush=True)
    assert llk >= prev - 1e-6 * abs(llk)
    prev = llk
