"""Allocation soak: repeated dataset / model / EM / mixture / output-pass cycles must not grow device memory (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import numpy as np
import ppca_rs_amd as P

rng = np.random.default_rng(0)
def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
base = None
for it in range(60):
    x = rng.standard_normal((20000, 96)); x[rng.random(x.shape) < 0.3] = np.nan
    ds = P.Dataset(x, rng.uniform(0.5, 1.5, 20000))
    m = P.PPCAModel.init(6, ds, seed=it)
    for _ in range(3):
        m = m.iterate(ds)
    m.llks(ds); m.extrapolate(ds); m.infer(ds).states()
    mix = P.PPCAMix.init(3, 4, ds, seed=it).iterate(ds)
    mix.smooth(ds); mix.infer_cluster(ds)
    big = P.PPCAModel.init(16, ds, seed=it).iterate(ds)   # generic path
    del ds, m, mix, big
    if it == 9: base = used()
    if it % 10 == 9: print(f"iteration {it+1}: {used():.0f} MiB in use", flush=True)
assert used() - base < 64, (base, used())
print("soak ok")
