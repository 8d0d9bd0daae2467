import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o
g = np.load(os.path.join(R, "tests/golden", sys.argv[1] if len(sys.argv) > 1 else "wide_d256_k10.npz"))
x, s, c, mu = g["x"], float(g["s0"]), g["c0"], g["mu0"]
m = P.PPCAModel(s, c, mu); ds = P.Dataset(x)
d, k = c.shape; kp = k * (k + 1) // 2
L = _lib.lib().ppca_stats_len(d, k)
got = np.empty(L); _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
want = o.stats(x, s, c, mu)
b = [0, d*k, d*k+d*kp, 2*d*k+d*kp, 2*d*k+d*kp+d, 2*d*k+d*kp+2*d, L]
for name, a, e in zip(["cross", "S", "U", "sumx", "totals", "scalars"], b[:-1], b[1:]):
    print(name, np.abs(got[a:e]-want[a:e]).max() / max(np.abs(want[a:e]).max(), 1e-300))
print("scalars got", got[-8:-3], "want", want[-8:-3])
tot_g, tot_w = got[b[4]:b[5]], want[b[4]:b[5]]
bad = np.nonzero(np.abs(tot_g - tot_w) > 1e-9)[0]
print("bad totals dims", bad[:20], (tot_g - tot_w)[bad[:20]])
