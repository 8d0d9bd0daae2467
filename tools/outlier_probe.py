import sys, numpy as np
sys.path.insert(0, "/root/repo")
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o
o.build()
ctx = _lib.default_context()
for k, d in ((10, 256), (16, 200)):
    n = 6000
    rng = np.random.default_rng(3)
    x, _, _ = o.synth(n, d, k, 0.3, 11)
    x[100] *= 1e6
    c, mu, s = 0.5 * rng.standard_normal((d, k)), np.zeros(d), 0.7
    m = P.PPCAModel(s, c, mu)
    L = _lib.lib().ppca_stats_len(d, k)
    want = o.stats(x, s, c, mu)
    kp = k * (k + 1) // 2
    for cap in (0, 1):
        ctx.set_grid_limit(cap)
        got = np.empty(L)
        ds = P.Dataset(x)
        _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ds._h, m._device(ctx).h, _lib.ptr(got)))
        S_g = got[d * k:d * k + d * kp].reshape(d, kp); S_w = want[d * k:d * k + d * kp].reshape(d, kp)
        masked = ~np.isfinite(x[100])
        # per-dimension relative error of the diagonal entries (sums of positive terms)
        diag = [a * (a + 1) // 2 + a for a in range(k)]
        rel = np.abs(S_g[:, diag] - S_w[:, diag]) / np.abs(S_w[:, diag])
        print(f"k={k} d={d} cap={cap}: S diag rel err: dims where the outlier is masked max {rel[masked].max():.3e}, observed max {rel[~masked].max():.3e};"
              f" block-max-relative {np.abs(S_g - S_w).max() / np.abs(S_w).max():.3e}; counters {ctx.debug_counters()}")
    ctx.set_grid_limit(0)
