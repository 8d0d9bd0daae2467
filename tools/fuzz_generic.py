"""Randomised parity sweep of the split pipeline (ppca_generic.hip) against the oracle: state sizes 11 .. 64, d up to 1200, chunk
sizes that force several chunks and K-split slices, both tile sizes of the int8 GEMM's ring loop (256-row tiles need >= 4096 rows
per chunk or d >= 1024), odd and even numbers of K-steps; every case also run twice for bit-identical statistics (a race in the
ring's handshake would show there first).  Diagnostic; run on the GPU box:  python tools/fuzz_generic.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o

if o.num_threads() > 16:
    o.set_threads(16)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))
worst, done = 0.0, 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    k = int(rng.integers(11, 65))
    flavour = case % 4
    if flavour == 0:    # many rows per chunk: the Gram product on 256-row tiles
        d, n = int(rng.integers(max(k, 257), 700)), int(rng.integers(4096, 9000))
    elif flavour == 1:  # d >= 1024: the statistics product on 256-row tiles
        d, n = int(rng.integers(1024, 1200)), int(rng.integers(1, 1500))
    else:
        d, n = int(rng.integers(max(k, 17), 600)), int(rng.integers(1, 3000))
    if k <= 16 and d <= 256:
        d = 300  # (11 <= k <= 16 at d <= 256 is the two-kernel pass, not this pipeline)
    while n * d * k * k > 4e9:
        n = max(1, n // 2)
    chunk = 64 * int(rng.integers(1, 9)) if (flavour >= 2 and rng.random() < 0.5) else 0
    if chunk: os.environ["PPCA_GEN_CHUNK"] = str(chunk)
    else: os.environ.pop("PPCA_GEN_CHUNK", None)
    x = rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.2 * rng.standard_normal((n, d)) + rng.standard_normal(d)
    kind = rng.integers(0, 3)
    if kind == 0: x[rng.random((n, d)) < rng.uniform(0, 0.8)] = np.nan
    elif kind == 1: x[rng.random(n) < 0.2] = np.nan; x[:, rng.random(d) < 0.1] = np.nan
    else:
        for i in range(n):
            st = rng.integers(0, d); x[i, (st + np.arange(d // 2)) % d] = np.nan
    if not np.isfinite(x).any(): continue
    w = rng.uniform(0.2, 2.0, n)
    c, mu, s = rng.standard_normal((d, k)) * rng.uniform(0.1, 1.0), rng.standard_normal(d), float(rng.uniform(0.1, 2.0))
    assert _lib.lib().ppca_path_kind(d, k) == 0
    ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
    L = _lib.lib().ppca_stats_len(d, k)
    got, again = np.empty(L), np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(again)))
    assert np.array_equal(got, again), (case, n, d, k, chunk, "not bit-reproducible")
    want = o.stats(x, s, c, mu, w)
    kp = k * (k + 1) // 2
    bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
    for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
        e = rel(got[a:b], want[a:b]); worst = max(worst, e)
        assert e < 1e-8, (case, n, d, k, chunk, kind, name, e)
    done += 1
    print("case", case, "n", n, "d", d, "k", k, "chunk", chunk or "auto", "guard", ds._ctx.last_guard(), flush=True)
os.environ.pop("PPCA_GEN_CHUNK", None)
print("fuzz ok:", done, "cases; worst relative deviation", worst)
