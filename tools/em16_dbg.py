import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o
for (n, d, k, mp) in [(90, 40, 12, 0.4), (3000, 128, 13, 0.3), (64, 40, 12, 0.4), (96, 40, 12, 0.0)]:
    rng = np.random.default_rng(n + d + k)
    x, _, _ = o.synth(n, d, k, mp, 900 + n + d + k)
    x[1] = np.nan
    w = rng.uniform(0.5, 1.5, n)
    c, mu, s = 0.3 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.9
    ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
    kp = k * (k + 1) // 2
    L = _lib.lib().ppca_stats_len(d, k)
    got = np.empty(L); _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    want = o.stats(x, s, c, mu, w)
    S_g = got[d*k:d*k+d*kp].reshape(d, kp); S_w = want[d*k:d*k+d*kp].reshape(d, kp)
    err = np.abs(S_g - S_w) / np.abs(S_w).max()
    idx = np.argsort(err.ravel())[::-1][:12]
    print("shape", n, d, k, "max", err.max())
    cols = sorted(set(int(i % kp) for i in idx)); print(" worst cols", cols, "dims", sorted(set(int(i // kp) for i in idx))[:12])
    print(" per-column max err:", " ".join("%d:%.0e" % (cc, err[:, cc].max()) for cc in range(kp) if err[:, cc].max() > 1e-12))
    print(" col abs max (want):", " ".join("%d:%.1e" % (cc, np.abs(S_w[:, cc]).max()) for cc in cols))
