"""Long-run soak of the EM kernels (diagnostic): many EM iterations at several sizes and grid caps -- the role-counter protocol
of the eight-wave kernels must never hang (every call is made under a watchdog by the caller: `timeout`), the llk must not
decrease, and two runs from the same start must agree bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib

ctx = _lib.default_context()
rng = np.random.default_rng(1)
t0 = time.time()
total = 0
for n, d, k, iters, cap in ((10_000_000, 256, 10, 120, 0), (3_000_000, 200, 7, 150, 0), (500_000, 256, 10, 200, 3), (64_123, 129, 4, 300, 1),
                            (2_000_000, 256, 10, 100, 17)):
    truth = P.PPCAModel(0.1, rng.standard_normal((d, k)), rng.standard_normal(d))
    ds = truth.sample(n, 0.3, seed=n % 1000)
    ctx.set_grid_limit(cap)
    runs = []
    for rep in range(2):
        m = P.PPCAModel.init(k, ds, seed=5)
        prev = -np.inf
        for it in range(iters):
            m, llk = m.iterate_with_llk(ds)
            assert np.isfinite(llk) and llk >= prev - 1e-9 * abs(llk), (n, d, k, it, llk, prev)
            prev = llk
        runs.append((m.isotropic_noise, m.transform.copy(), prev))
        total += iters
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2], (n, d, k, "not reproducible")
    print(f"N={n} d={d} k={k} cap={cap}: {iters} iterations x 2, llk/N {prev / n:.6f}, sigma {m.isotropic_noise:.6f}, guards {ctx.last_guard()}", flush=True)
ctx.set_grid_limit(0)
print(f"soak ok: {total} EM iterations in {time.time() - t0:.0f} s")
