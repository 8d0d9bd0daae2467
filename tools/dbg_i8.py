import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ppca_rs_amd as P
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/wide_d256_k10.npz"))
x, s, c, mu = g["x"], float(g["s0"]), g["c0"], g["mu0"]
m = P.PPCAModel(s, c, mu); ds = P.Dataset(x)
l = m.llks(ds)
err = np.abs(l - g["llks"]) / np.abs(g["llks"]).max()
print("llks rel err max", err.max(), "argmax", err.argmax(), "first errs", err[:8])
inf = m.infer(ds)
cv = np.array(inf.covariances())
print("cov rel err", np.abs(cv - g["covs"]).max() / np.abs(g["covs"]).max(), "state err", np.abs(inf.states() - g["states"]).max())
# per-entry error pattern of Sigma for sample 0
d0 = np.abs(cv[0] - g["covs"][0]); print(np.round(np.log10(d0 + 1e-30)).astype(int))
