"""Randomised parity sweep aimed at the fixed-point statistics (round 4): outlier rows, heavy-tailed weights, capped grids
(a workgroup walks many tiles), copies of one row (correlated roundings: round 5), every engine (eight-wave kernel k <= 10, two-kernel pass k = 11..16, split pipeline) --
the statistics against the oracle block by block AND, for the diagonal of S and the totals, DIMENSION BY DIMENSION
(element-wise relative), which is what an outlier row breaks when the guard is missing.  Diagnostic; run on the GPU box:

    python tools/fuzz_gpu2.py [seed] [cases]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o

o.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
ctx = _lib.default_context()
worst_block, worst_elem = 0.0, 0.0
tally = {"fused": 0, "em16": 0, "generic": 0, "guard_w": 0, "fallback16": 0}


def one_case(case):
    global worst_block, worst_elem
    engine = ("fused", "em16", "generic")[case % 3]
    if engine == "fused":
        k = int(rng.integers(1, 11)); d = int(rng.integers(max(k, 2), 257))
    elif engine == "em16":
        k = int(rng.integers(11, 17)); d = int(rng.integers(k, 257))
    else:
        k = int(rng.integers(1, 25)); d = int(rng.integers(max(257, k), 420)) if k <= 16 else int(rng.integers(k, 200))
    n = int(rng.integers(200, 3000))
    x = rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.1 * rng.standard_normal((n, d)) + rng.standard_normal(d)
    x[rng.random((n, d)) < rng.uniform(0.05, 0.6)] = np.nan
    flavour = int(rng.integers(0, 6))
    w = None
    if flavour in (4, 5):
        # CORRELATED roundings (round 5): most rows are copies of ONE row (same values, same mask), so the cut of their [wP | wz | w]
        # rows rounds every copy the same way -- the guard's bound counts 4 sqrt(rows) quanta per flush window as if the roundings were
        # independent; with copies the error of a column sum grows like rows x quantum / 2.  An outlier row among them lifts the
        # exponents (flavour 5: and masks some dimensions), which is where coarse cuts + correlated roundings could add up.
        src = int(rng.integers(0, n))
        copies = rng.random(n) < 0.9
        x[copies] = x[src]
        big = int(rng.integers(0, n))
        x[big] = rng.standard_normal(d) * 10.0 ** rng.uniform(3, 7)
        if flavour == 5:
            x[big, rng.random(d) < 0.5] = np.nan
    if flavour in (1, 3):  # outlier rows
        for i in rng.choice(n, size=int(rng.integers(1, 4)), replace=False):
            x[i] *= 10.0 ** rng.uniform(2, 8)
    if flavour in (2, 3):  # heavy-tailed weights
        w = np.exp(rng.standard_normal(n) * rng.uniform(2, 12))
        w /= w.max()
        w[rng.random(n) < 0.05] = 0.0
    c, mu, s = rng.standard_normal((d, k)) * rng.uniform(0.2, 2), rng.standard_normal(d) * 0.3, float(rng.uniform(0.2, 2.0))
    cap = int(rng.choice([0, 1, 2]))
    ctx.set_grid_limit(cap)
    ctx.debug_counters(reset=True)
    ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
    L = _lib.lib().ppca_stats_len(d, k)
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ds._h, m._device(ctx).h, _lib.ptr(got)))
    want = o.stats(x, s, c, mu, w)
    kp = k * (k + 1) // 2
    b = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
    for name, lo, hi in zip(["cross", "S", "U", "sumx", "totals", "scalars"], b[:-1], b[1:]):
        e = float(np.abs(got[lo:hi] - want[lo:hi]).max() / max(np.abs(want[lo:hi]).max(), 1e-300))
        worst_block = max(worst_block, e)
        assert e < 1e-8, (case, engine, n, d, k, flavour, cap, name, e)
    diag = [a * (a + 1) // 2 + a for a in range(k)]
    Sg, Sw = got[b[1]:b[2]].reshape(d, kp)[:, diag], want[b[1]:b[2]].reshape(d, kp)[:, diag]
    tg, tw = got[b[4]:b[5]], want[b[4]:b[5]]
    live = tw > 0
    if live.any():
        e = float((np.abs(Sg[live] - Sw[live]) / np.abs(Sw[live])).max())
        e = max(e, float((np.abs(tg[live] - tw[live]) / tw[live]).max()))
        worst_elem = max(worst_elem, e)
        assert e < 1e-7, (case, engine, n, d, k, flavour, cap, "element-wise S diagonal / totals", e)
    tally[engine] += 1
    if engine == "fused":
        tally["guard_w"] += ctx.last_guard()[1]
    tally["fallback16"] += 1 if ctx.debug_counters()[7] else 0


for case in range(cases):
    one_case(case)
ctx.set_grid_limit(0)
print("fuzz2 ok: seed %d, %d cases %s; worst block-relative %.2e, worst element-wise (S diagonal, totals) %.2e" % (seed, cases, tally, worst_block, worst_elem))
