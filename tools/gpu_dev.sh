mkdir -p gpurun_out/r2t
export PPCA_HIP_LIB=$PWD/ppca_rs_amd/libppca_hip_dev.so
timeout 600 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o
o.build()
ctx = _lib.default_context()
rng = np.random.default_rng(5)
for n in (1, 31, 32, 33, 64, 65, 97, 1000, 4097):
    x, _, _ = o.synth(n, 256, 10, 0.3, 77 + n)
    if n > 10: x[3] = np.nan
    c, mu, s = rng.standard_normal((256, 10)), 0.1 * rng.standard_normal(256), 0.7
    w = rng.uniform(0.5, 1.5, n)
    ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
    got, want = m.llks(ds), o.llks(x, s, c, mu)
    rel = np.abs(got - want).max() / np.abs(want).max()
    tot, wtot = m.llk(ds), o.llk(x, s, c, mu, w)
    print(n, 'llks rel', rel, 'llk rel', abs(tot - wtot) / abs(wtot))
    assert rel < 1e-10 and abs(tot - wtot) < 1e-10 * abs(wtot)
# d < 256 and timing
x, _, _ = o.synth(5000, 200, 10, 0.4, 9)
c, mu, s = rng.standard_normal((200, 10)), 0.1 * rng.standard_normal(200), 0.9
assert np.abs(P.PPCAModel(s, c, mu).llks(P.Dataset(x)) - o.llks(x, s, c, mu)).max() < 1e-9 * 1e3
truth = P.PPCAModel(0.1, rng.standard_normal((256, 10)), rng.standard_normal(256))
ds = truth.sample(4_000_000, 0.3, seed=1)
m = P.PPCAModel.init(10, ds, seed=2)
import ctypes as C
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    v = m.llk(ds)
    ctx.synchronize(); print('llk pass N=4M: %.2f ms' % ((time.perf_counter() - t0) * 1e3), v / 4e6)
PY
PPCA_LLK2=0 timeout 300 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(5)
truth = P.PPCAModel(0.1, rng.standard_normal((256, 10)), rng.standard_normal(256))
ds = truth.sample(4_000_000, 0.3, seed=1)
m = P.PPCAModel.init(10, ds, seed=2)
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    v = m.llk(ds)
    ctx.synchronize(); print('old llk pass N=4M: %.2f ms' % ((time.perf_counter() - t0) * 1e3), v / 4e6)
PY
