import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ppca_rs_amd as P
x = np.random.default_rng(0).standard_normal((1_000_000, 256))
x[x > 2.5] = np.nan
P.Dataset(x[:1000])
t0 = time.perf_counter(); ds = P.Dataset(x); ds._ctx.synchronize(); t1 = time.perf_counter()
print(f"upload {x.nbytes/1e9:.2f} GB in {t1-t0:.3f} s = {x.nbytes/1e9/(t1-t0):.1f} GB/s")
t0 = time.perf_counter(); y = ds.numpy(); t1 = time.perf_counter()
print(f"download in {t1-t0:.3f} s = {x.nbytes/1e9/(t1-t0):.1f} GB/s")
