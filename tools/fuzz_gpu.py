"""Randomised parity sweep of the fused path against the oracle: shapes, mask patterns (iid, whole rows, whole dims,
runs), weights including zeros, extreme-but-sane scales (diagnostic; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    k = int(os.environ.get('PPCA_FUZZ_K', 0)) or int(rng.integers(1, 11)); d = int(rng.integers(k, 257)); n = int(rng.integers(1, 400))
    if case % 5 == 0: d = 256
    x = rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.1 * rng.standard_normal((n, d)) + rng.standard_normal(d)
    kind = rng.integers(0, 4)
    if kind == 0: x[rng.random((n, d)) < rng.uniform(0, 0.9)] = np.nan
    elif kind == 1: x[rng.random(n) < 0.3] = np.nan; x[:, rng.random(d) < 0.2] = np.nan
    elif kind == 2:
        for i in range(n):
            st = rng.integers(0, d); x[i, (st + np.arange(d // 2)) % d] = np.nan
    x[rng.random((n, d)) < 0.01] = np.inf
    w = rng.uniform(0.0, 2.0, n); w[rng.random(n) < 0.1] = 0.0
    if not (w.sum() > 0): w[0] = 1.0
    c, mu, s = rng.standard_normal((d, k)) * rng.uniform(0.1, 3), rng.standard_normal(d), float(rng.uniform(0.05, 3.0))
    xn = np.where(np.isfinite(x), x, np.nan)
    if not np.isfinite(xn).any(): continue
    for ww in (w, None):
        ds = P.Dataset(x, ww); m = P.PPCAModel(s, c, mu)
        got = np.empty(_lib.lib().ppca_stats_len(d, k))
        _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
        want = o.stats(xn, s, c, mu, ww)
        e = rel(got, want); worst = max(worst, e)
        assert e < 1e-8, (case, n, d, k, kind, ww is None, e)
    e = rel(m.llks(ds), o.llks(xn, s, c, mu)); worst = max(worst, e); assert e < 1e-9, (case, "llks", e)
    e = rel(m.extrapolate(ds).numpy(), o.reconstruct(xn, s, c, mu, "extrapolate")); worst = max(worst, e); assert e < 1e-8, (case, "extrapolate", e)
print("fuzz ok; worst relative deviation", worst)
