"""Shard additivity of the statistics at the headline size, per block (diagnostic): N rows generated on the device,
a model three EM steps from init; full pass vs the sum of 8 row shards."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, ".")
import ppca_rs_amd as P
from ppca_rs_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d, k = 256, 10
ctx = _lib.default_context()
rng = np.random.default_rng(1)
truth = P.PPCAModel(0.1, rng.standard_normal((d, k)), rng.standard_normal(d))
ds = truth.sample(n, 0.3, seed=5)
m = P.PPCAModel.init(k, ds, seed=11)
for _ in range(3):
    m = m.iterate(ds)
L = _lib.lib().ppca_stats_len(d, k)
def stats(x):
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, x._h, m._device(ctx).h, _lib.ptr(got)))
    return got
full = stats(ds)
again = stats(ds)
acc = np.zeros(L)
for ch in ds.chunks(8):
    acc += stats(ch)
kp = k * (k + 1) // 2
b = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
print("lib", _lib.LIB_PATH.split("/")[-1], "n", n, "repro", np.array_equal(full, again))
for name, lo, hi in zip(["cross", "S", "U", "sumx", "totals", "scalars"], b[:-1], b[1:]):
    dif = np.abs(acc[lo:hi] - full[lo:hi])
    print("  %-8s max|diff| %.3e  at %d  rel-to-block-max %.3e  worst elementwise rel %.3e" % (
        name, dif.max(), dif.argmax(), dif.max() / np.abs(full[lo:hi]).max(), (dif / np.maximum(np.abs(full[lo:hi]), 1e-300)).max()))
print("  scalars full", full[-8:-3], "\n  scalars acc ", acc[-8:-3])
