"""Parity of the pipelined-solve variant (PPCA_EM9=1) against the oracle: N = 20 000 on the full grid and on 2 / 1 workgroups,
weighted and not, plus ragged shapes (diagnostic; tools/devbuild.py libraries hold k = 10 only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o
o.build()
ctx = _lib.default_context()
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
worst = 0.0
for (n, d, k) in ((20000, 256, 10), (97, 256, 10), (33, 200, 10), (4001, 255, 10), (1, 17, 10), (64, 64, 10)):
    rng = np.random.default_rng(n)
    x, _, _ = o.synth(n, d, k, 0.3, 7 + n)
    if n > 40: x[n // 3] = np.nan
    c, mu, s = 0.6 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), 0.7
    w = rng.uniform(0.25, 2.0, n)
    m = P.PPCAModel(s, c, mu)
    L = _lib.lib().ppca_stats_len(d, k)
    kp = k * (k + 1) // 2
    b = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
    for cap in (0, 2, 1):
        ctx.set_grid_limit(cap)
        for ww in (None, w):
            got = np.empty(L)
            ds = P.Dataset(x, ww)
            _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ds._h, m._device(ctx).h, _lib.ptr(got)))
            want = o.stats(x, s, c, mu, ww)
            for name, lo, hi in zip(["cross", "S", "U", "sumx", "totals", "scalars"], b[:-1], b[1:]):
                e = rel(got[lo:hi], want[lo:hi]); worst = max(worst, e)
                assert e < 1e-9, (n, d, cap, ww is None, name, e)
ctx.set_grid_limit(0)
print("em9 check ok (PPCA_EM9=%s): worst block-relative deviation %.2e" % (os.environ.get("PPCA_EM9"), worst))
