"""Parity of one EM step and the llk against the oracle at awkward scales (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from oracle import ppca_oracle as o

rng = np.random.default_rng(0)
for d, k in ((256, 10), (40, 6)):
    for scale, sigma in ((1.0, 1.0), (1e4, 1e-3), (1e-4, 1e-6), (1e6, 10.0), (1.0, 1e-5)):
        x, _, _ = o.synth(3000, d, k, 0.3, 7)
        x = x * scale
        x[5] = np.nan
        c = rng.standard_normal((d, k)) * scale
        mu = rng.standard_normal(d) * scale
        ds, m = P.Dataset(x), P.PPCAModel(sigma, c, mu)
        new, llk = m.iterate_with_llk(ds)
        s1, c1, m1 = o.iterate(x, sigma, c, mu)
        want = o.llk(x, sigma, c, mu)
        rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
        print(f"d={d} k={k} scale={scale:g} sigma={sigma:g}: llk rel {abs(llk-want)/abs(want):.2e}  sigma' rel {abs(new.isotropic_noise-s1)/s1:.2e}  "
              f"C' rel {rel(new.transform, c1):.2e}  mean' rel {rel(new.mean, m1):.2e}  llks rel {rel(m.llks(ds), o.llks(x, sigma, c, mu)):.2e}", flush=True)

print("EM monotonicity where the literal form breaks down (scale 1e4, sigma 1e-3): llk of the input model, of our update, of the oracle's update")
x, _, _ = o.synth(3000, 256, 10, 0.3, 7)
x = x * 1e4
c = np.random.default_rng(1).standard_normal((256, 10)) * 1e4
mu = np.zeros(256)
ds, m = P.Dataset(x), P.PPCAModel(1e-3, c, mu)
for it in range(3):
    new = m.iterate(ds)
    s1, c1, m1 = o.iterate(x, m.isotropic_noise, m.transform, m.mean)
    print(f"  iter {it}: input {m.llk(ds):.6e}   ours {new.llk(ds):.6e}   oracle's {P.PPCAModel(s1, c1, m1).llk(ds):.6e}", flush=True)
    m = new
