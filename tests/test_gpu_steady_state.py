"""The STEADY STATE of the fused kernels under the oracle.

The fused kernels start min(tiles, CUs) persistent workgroups, so at the sizes the CPU oracle affords (N <= 20 000) a
workgroup of the 256-CU part walks at most three 32-sample tiles -- what only exists from tile 2-4 onwards (em8's
three-tile mask ring and its two-blocks-deep pipelined contraction, the periodic flush of the int64 accumulators every
100 groups, a raise of the fixed-point exponents with a contraction pending; em16's next-tile prefetch, LDS-parked factor
and hand-over ring; the two-tile rounds of the llk sweep) would only ever be compared with itself.  Here:

  * `ppca_ctx_set_grid_limit` (a test hook of the C-ABI) caps the grid at 1-2 workgroups, so N = 20 000 gives every
    workgroup ~300 tiles, and every pass is compared with the literal oracle (ppca_model.rs:277-358 et al.);
  * weights spanning 2^80 INSIDE one workgroup's run force the rescale path again and again -- asserted through the
    kernels' diagnostic counters (`ppca_debug_counters`), not assumed;
  * BASELINE config 2 at its full size (N = 1 M) against the oracle's one-sweep form (itself pinned to the literal port
    in tests/test_oracle.py) and 200 000 rows against the literal port; the reference's largest shape (d = 200, k = 16)
    likewise;
  * BASELINE config 3's size (N = 10 M on one GPU: the periodic flush fires six times per workgroup) through
    size-independent properties: 8-shard additivity, bit reproducibility, monotone llk, observed counts BIT-EXACT.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(hiplib):
    import ppca_rs_amd as p

    return p


@pytest.fixture()
def ctx(P):
    from ppca_rs_amd import _lib

    c = _lib.default_context()
    c.set_grid_limit(0)
    c.set_heavy_rows(8)
    c.debug_counters(reset=True)
    yield c
    c.set_grid_limit(0)
    c.set_heavy_rows(8)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _blocks(d, k):
    kp = k * (k + 1) // 2
    b = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d]
    return list(zip(["cross", "S", "U", "sumx", "totals", "scalars"], b, b[1:] + [b[-1] + 8]))


def _stats(P, ds, m):
    from ppca_rs_amd import _lib

    got = np.empty(_lib.lib().ppca_stats_len(m.output_size, m.state_size))
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    return got


def _assert_stats(got, want, d, k, tol, tag):
    for name, a, b in _blocks(d, k):
        assert _rel(got[a:b], want[a:b]) < tol, (name,) + tuple(tag)


def _gathered_stats(P, ctx, ds, m, u):
    """One component pass of the mixture (ppca_mix_component_stats): weights exp(u - max u), rows of negligible weight
    dropped and the others GATHERED -- em8_kernel<K, true, true>."""
    import torch
    from ppca_rs_amd import _lib

    L = _lib.lib().ppca_stats_len(m.output_size, m.state_size)
    ud = torch.from_numpy(u).cuda()
    out = torch.empty(L, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    sm, used = C.c_double(0.0), C.c_int64(0)
    _lib.check(_lib.lib().ppca_mix_component_stats(ctx.handle, ds._h, m._device(ctx).h, C.c_void_p(ud.data_ptr()), float(u.max()),
                                                   C.c_void_p(out.data_ptr()), C.byref(sm), C.byref(used)))
    return out.cpu().numpy(), sm.value, used.value


@pytest.mark.parametrize("k,d", [(1, 64), (4, 200), (7, 255), (10, 256)])
def test_eight_wave_em_pass_steady_state(P, oracle, ctx, k, d):
    """em8_kernel (weighted, un-weighted and gathered instantiations) with ~300 tiles per workgroup: the three-tile mask
    ring, the pipelined contraction, the tiles-cut counter and (beyond 200 tiles) the periodic int64 flush all fire; the
    whole statistics buffer against the literal oracle."""
    n = 20_000
    rng = np.random.default_rng(500 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 7000 + k)
    x[n // 3] = np.nan  # an all-masked row in the middle of a run
    c, mu, s = 0.6 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), 0.7
    w = rng.uniform(0.25, 2.0, n)
    m = P.PPCAModel(s, c, mu)
    for cap in (2, 1):
        ctx.set_grid_limit(cap)
        for weights in (None, w):
            ctx.debug_counters(reset=True)
            got = _stats(P, P.Dataset(x, weights), m)
            cnt = ctx.debug_counters()
            assert cnt[2] >= (n // 32) // cap and cnt[1] >= 1, cnt  # tiles walked by one workgroup; periodic flushes
            assert ctx.last_guard() == (0, 0)  # (the int8 engine on both sides: neither guard sent the pass to fp64)
            _assert_stats(got, oracle.stats(x, s, c, mu, weights), d, k, 1e-9, (k, cap, weights is None))
        if cap == 2:
            # gathered: a third of the rows dropped (weight exactly 0), the others with weights over 20 binary orders
            u = np.log(w) + rng.uniform(-14.0, 0.0, n)
            u[rng.random(n) < 0.33] = -np.inf
            got, sm, used = _gathered_stats(P, ctx, P.Dataset(x), m, u)
            wg = np.exp(u - u.max())
            assert used == int((wg > 0).sum()) and used // 32 // cap > 150
            assert abs(sm - wg.sum()) < 1e-12 * wg.sum()
            want = oracle.stats(x, s, c, mu, wg)
            for name, a, b in _blocks(d, k)[:-1]:
                assert _rel(got[a:b], want[a:b]) < 1e-9, (name, k, "gathered")
            # scalars: a component pass skips the log-likelihood (PassArgs::no_llk) and counts the gathered rows only
            assert _rel(got[[-8, -7, -5]], want[[-8, -7, -5]]) < 1e-9


@pytest.mark.parametrize("k,d", [(1, 64), (4, 200), (7, 255), (10, 256)])
def test_output_passes_steady_state(P, oracle, ctx, k, d):
    """llk2_kernel (two-tile rounds) and pass_kernel<K, false> (states, covariances, smooth, extrapolate, covariance
    diagonals) with ~300 tiles per workgroup against the oracle."""
    n = 20_000
    rng = np.random.default_rng(600 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 7100 + k)
    x[n // 2] = np.nan
    c, mu, s = 0.6 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), 0.7
    w = rng.uniform(0.25, 2.0, n)
    m = P.PPCAModel(s, c, mu)
    ds = P.Dataset(x, w)
    ctx.set_grid_limit(2)
    assert _rel(m.llks(ds), oracle.llks(x, s, c, mu)) < 1e-10
    want = oracle.llk(x, s, c, mu, w)
    assert abs(m.llk(ds) - want) < 1e-10 * abs(want)
    st, cv = oracle.infer(x, s, c, mu)
    inf = m.infer(ds)
    assert _rel(inf.states(), st) < 1e-9 and _rel(np.array(inf.covariances()), cv) < 1e-9
    assert _rel(m.smooth(ds).numpy(), oracle.reconstruct(x, s, c, mu, "smooth")) < 1e-9
    ex = m.extrapolate(ds).numpy()
    assert _rel(ex, oracle.reconstruct(x, s, c, mu, "extrapolate")) < 1e-9
    assert np.array_equal(ex[np.isfinite(x)], x[np.isfinite(x)])
    from ppca_rs_amd import _lib

    for mode, name in ((0, "smooth"), (1, "extrapolate")):
        h = C.c_void_p()
        _lib.check(_lib.lib().ppca_covariance_diagonal(ctx.handle, ds._h, m._device(ctx).h, mode, C.byref(h)))
        assert _rel(P.Dataset._wrap(h, ctx).numpy(), oracle.covariance_diagonal(x, s, c, mu, name)) < 1e-9


@pytest.mark.parametrize("k,d", [(11, 256), (13, 200), (14, 255), (16, 200)])
def test_two_kernel_em_pass_steady_state(P, oracle, ctx, k, d):
    """estep16_kernel + sstat16_kernel with ~300 tiles per workgroup (next-tile prefetch, LDS-parked factor, hand-over
    ring, flush cadence), weighted and un-weighted, and the output passes of these shapes, against the oracle."""
    n = 20_000
    rng = np.random.default_rng(700 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 7200 + k)
    x[n // 3] = np.nan
    c, mu, s = 0.5 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), 0.7
    w = rng.uniform(0.25, 2.0, n)
    m = P.PPCAModel(s, c, mu)
    for cap in (2, 1):
        ctx.set_grid_limit(cap)
        for weights in (None, w):
            ctx.debug_counters(reset=True)
            got = _stats(P, P.Dataset(x, weights), m)
            cnt = ctx.debug_counters()
            assert cnt[6] >= (n // 32) // cap and cnt[5] >= 1 and cnt[7] == 0, cnt  # (no workgroup needed the fp64 statistics)
            _assert_stats(got, oracle.stats(x, s, c, mu, weights), d, k, 1e-9, (k, cap, weights is None))
    ctx.set_grid_limit(2)
    ds = P.Dataset(x, w)
    assert _rel(m.llks(ds), oracle.llks(x, s, c, mu)) < 1e-10
    st, cv = oracle.infer(x, s, c, mu)
    inf = m.infer(ds)
    assert _rel(inf.states(), st) < 1e-9 and _rel(np.array(inf.covariances()), cv) < 1e-9
    assert _rel(m.smooth(ds).numpy(), oracle.reconstruct(x, s, c, mu, "smooth")) < 1e-9
    assert _rel(m.extrapolate(ds).numpy(), oracle.reconstruct(x, s, c, mu, "extrapolate")) < 1e-9
    assert _rel(inf.smoothed_covariances_diagonal(m).numpy(), oracle.covariance_diagonal(x, s, c, mu, "smooth")) < 1e-9


@pytest.mark.parametrize("k,d", [(4, 64), (10, 256), (12, 64), (16, 200)])
def test_fixed_point_form_rescales_inside_one_workgroup(P, oracle, ctx, k, d):
    """Weights spanning 2^80 over ONE workgroup's rows (grid capped at 1): ascending, a column outgrows its exponent every
    few tiles -- contract what is pending under the old exponents, flush the int64 accumulators into the partial, raise
    the exponents, cut the tile again --, asserted through the rescale counter; descending, the exponents stay where the
    first tile put them and the later rows are cut far below it (precision relative to the LARGEST terms, as in any fp64
    sum).  em8_kernel's back role (k <= 10) and sstat16_kernel (k >= 11), against the oracle."""
    n = 6_000
    rng = np.random.default_rng(800 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 7300 + k)
    c, mu, s = 0.4 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.8
    m = P.PPCAModel(s, c, mu)
    ctx.set_grid_limit(1)
    base = 0 if k <= 10 else 4
    for asc in (True, False):
        w = 2.0 ** np.linspace(-40, 40, n)
        w = w if asc else w[::-1].copy()
        ctx.debug_counters(reset=True)
        got = _stats(P, P.Dataset(x, w), m)
        cnt = ctx.debug_counters()
        assert cnt[base + 2] == (n + 31) // 32
        if asc:
            assert cnt[base] >= 8, cnt  # 80 binary orders at 6 orders of head room per rescale
        else:
            assert cnt[base] == 0, cnt
        _assert_stats(got, oracle.stats(x, s, c, mu, w), d, k, 1e-9, (k, asc))


def _generated(P, ctx, n, d, k, seed, mask_prob=0.3):
    from ppca_rs_amd import _lib

    truth = P.PPCAModel(0.1, np.random.default_rng(seed).standard_normal((d, k)), np.random.default_rng(seed + 1).standard_normal(d))
    spec = _lib.SynthSpec(0, n, d, k, 0.1, mask_prob, 0, 0, 1000 + seed, truth._c.ctypes.data_as(_lib.c_double_p),
                          truth._mean.ctypes.data_as(_lib.c_double_p))
    h = C.c_void_p()
    _lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
    return P.Dataset._wrap(h, ctx)


@pytest.mark.parametrize("d,k", [(256, 10), (200, 16)])
def test_full_size_statistics_against_oracle(P, oracle, ctx, d, k):
    """BASELINE config 2 at its own size (N = 1 M, d = 256, k = 10, 30 % masked; 122 tiles per workgroup on the full
    grid) and the reference's largest shape (d = 200, k = 16, lib.rs:82-99): the statistics buffer of one EM pass
    against the oracle's one-sweep form on all rows, and against the LITERAL port (four sweeps, ppca_model.rs:277-358)
    on the first 200 000."""
    n = 1_000_000
    ds = _generated(P, ctx, n, d, k, 31 + k)
    m = P.PPCAModel.init(k, ds, seed=5)
    m = m.iterate(ds)  # (a model one step away from the random start)
    s, c, mu = m.isotropic_noise, m.transform, m.mean
    x = ds.numpy()
    got = _stats(P, ds, m)
    _assert_stats(got, oracle.fused_stats(x, s, c, mu), d, k, 1e-9, (d, k, "one-sweep"))
    sub = ds._slice(0, 200_000)
    _assert_stats(_stats(P, sub, m), oracle.stats(x[:200_000], s, c, mu), d, k, 1e-9, (d, k, "literal"))


def test_full_size_properties_config3(P, ctx):
    """BASELINE config 3's size on one GPU (N = 10 M, d = 256, k = 10, 30 % masked: 1 221 tiles per workgroup, the
    periodic flush fires six times in each): 8-shard additivity (the multi-GPU invariant), run-to-run bit
    reproducibility, monotone llk, and the per-dimension observed counts -- sums of 0/1 through the int8 / int64 path --
    BIT-EXACT against a scan of the data by torch."""
    import torch
    from ppca_rs_amd import _lib

    n, d, k = 10_000_000, 256, 10
    g = torch.Generator(device="cuda").manual_seed(3)
    ct = torch.randn(d, k, dtype=torch.float64, device="cuda", generator=g)
    X = torch.empty(n, d, dtype=torch.float64, device="cuda")
    step = 1_000_000
    for a in range(0, n, step):
        z = torch.randn(step, k, dtype=torch.float64, device="cuda", generator=g)
        blk = z @ ct.T + 0.1 * torch.randn(step, d, dtype=torch.float64, device="cuda", generator=g)
        blk[torch.rand(step, d, device="cuda", generator=g) < 0.3] = float("nan")
        X[a:a + step] = blk
    del z, blk
    counts = torch.isfinite(X).sum(dim=0).double().cpu().numpy()
    torch.cuda.synchronize()
    ds = P.Dataset.from_device(X.data_ptr(), n, d, ctx=ctx, keepalive=X)
    m = P.PPCAModel.init(k, ds, seed=11)
    prev = -np.inf
    for _ in range(3):
        m, llk = m.iterate_with_llk(ds)
        assert np.isfinite(llk) and llk >= prev
        prev = llk
    ctx.debug_counters(reset=True)
    full = _stats(P, ds, m)
    cnt = ctx.debug_counters()
    assert cnt[1] >= 6 * 255 and cnt[2] == 1221, cnt  # (the last workgroup's run is shorter)
    np.testing.assert_array_equal(full, _stats(P, ds, m))
    L = len(full)
    o_tot = 2 * d * k + d * (k * (k + 1) // 2) + d
    np.testing.assert_array_equal(full[o_tot:o_tot + d], counts)
    acc = np.zeros(L)
    for ch in ds.chunks(8):
        acc += _stats(P, ch, m)
    assert _rel(acc, full) < 1e-11
    np.testing.assert_array_equal(acc[o_tot:o_tot + d], counts)


def test_sharded_mixture_hands_out_detached_models(P, ctx):
    """ShardedMixEM ping-pongs between two private device models per component; a mixture handed out mid-training must
    not alias them (two steps later the buffer holds another iterate): its llk stays what it was."""
    from ppca_rs_amd.distributed import ShardedMixEM

    rng = np.random.default_rng(9)
    d, k, nm, n = 24, 3, 3, 3000
    parts = [P.PPCAModel(0.2, rng.standard_normal((d, k)), 3.0 * rng.standard_normal(d)).sample(n // nm, 0.2, seed=40 + c) for c in range(nm)]
    ds = P.Dataset.concat(parts)
    em = ShardedMixEM(ds, P.PPCAMix.init(nm, k, ds, seed=2))
    em.step()
    mid = em.mixture()
    before = mid.llk(ds)
    host = [np.array(m.transform) for m in mid.models]
    for _ in range(3):
        em.step()
    assert mid.llk(ds) == before
    for m, c in zip(mid.models, host):
        np.testing.assert_array_equal(m.transform, c)
    assert em.mixture().llk(ds) > before


@pytest.mark.parametrize("k,d", [(10, 256), (4, 64), (16, 200), (12, 64)])
def test_outlier_row_sends_the_statistics_to_fp64(P, oracle, ctx, k, d):
    """One row at 1e6 x the others lifts the column exponents of its workgroup's fixed-point form; the dimensions MASKED in
    that row then sum rows cut far below their own resolution (measured before the guard: S_j off by 4e-5 on the full
    grid, 1e-3 .. 1e-2 on one workgroup).  wguard_kernel sees it in the reduced statistics (a diagonal entry of S not
    large against the rounding bound of its column) and the pass is repeated with fp64 accumulation, as the reference sums
    (ppca_model.rs:297-306): every block within 1e-9 of the oracle, dimension by dimension where the large row is masked;
    the same data without the outlier stays on the int8 engine.  (Round 6: the eight-wave kernel now sends such a row ROUND its
    fixed-point form -- test_outlier_rows_go_round_the_fixed_point_form; ppca_ctx_set_heavy_rows(0) restores the behaviour this test
    is about, the guard behind it.)"""
    ctx.set_heavy_rows(0)
    n = 6000
    rng = np.random.default_rng(3)
    x, _, _ = oracle.synth(n, d, k, 0.3, 11)
    c, mu, s = 0.5 * rng.standard_normal((d, k)), np.zeros(d), 0.7
    m = P.PPCAModel(s, c, mu)
    kp = k * (k + 1) // 2
    diag = [a * (a + 1) // 2 + a for a in range(k)]
    for cap in (0, 1):
        ctx.set_grid_limit(cap)
        ctx.debug_counters(reset=True)
        _assert_stats(_stats(P, P.Dataset(x), m), oracle.stats(x, s, c, mu), d, k, 1e-9, (k, cap, "clean"))
        # k <= 10: the global check behind wguard_kernel; k = 11..16: sstat16_kernel's per-workgroup check (counter 7)
        assert (ctx.last_guard() == (0, 0)) if k <= 10 else (ctx.debug_counters()[7] == 0)
        xo = x.copy()
        xo[100] *= 1e6
        got, want = _stats(P, P.Dataset(xo), m), oracle.stats(xo, s, c, mu)
        assert (ctx.last_guard() == (0, 1)) if k <= 10 else (ctx.debug_counters()[7] >= 1), cap
        if k <= 10:
            # full grid (188 workgroups of one tile): only the outlier's workgroup is recomputed -- 32 rows; one workgroup: it IS
            # the whole pass
            mode, wgs, rows, _ = ctx.last_fallback()
            assert (mode, wgs, rows) == ((2, 1, 32) if cap == 0 else (1, 1, 0)), (cap, mode, wgs, rows)
        _assert_stats(got, want, d, k, 1e-9, (k, cap, "outlier"))
        Sg, Sw = got[d * k:d * k + d * kp].reshape(d, kp), want[d * k:d * k + d * kp].reshape(d, kp)
        masked = ~np.isfinite(xo[100])
        assert masked.any() and (np.abs(Sg[masked][:, diag] - Sw[masked][:, diag]) / np.abs(Sw[masked][:, diag])).max() < 1e-9
        # (an EM step from here solves row systems of condition ~1e12 -- the outlier dominates every S_j it is observed in --,
        #  so 1e-9 in the statistics is all two fp64 implementations can agree to; the noise level, a plain sum, still does)
        s1, _, _ = oracle.iterate(xo, s, c, mu)
        new = m.iterate(P.Dataset(xo))
        assert abs(new.isotropic_noise - s1) < 1e-9 * s1 and np.isfinite(new.transform).all()


@pytest.mark.parametrize("k", [4, 10])
def test_outlier_rows_cost_their_workgroups_slices(P, oracle, ctx, k):
    """The second stage of the guarded EM pass (round 5; ppca_kernels.hip, reduce_wguard_kernel): the last workgroup of the
    reduction flags the workgroups whose cut dominates the rounding bound, writes their slices out as a gather list, and the
    fp64 instantiation of the pass walks that list with the whole grid; the statistics are re-reduced from the un-flagged
    partials + the fallback's.  N = 20 000 on 8 workgroups (79 tiles each) with three outlier rows in two of them: mode 2,
    two workgroups, 2 x 79 x 32 rows, every block within 1e-9 of the literal oracle (ppca_model.rs:277-358) -- un-weighted,
    weighted (the list carries the rows' weights) and as a GATHERED component pass of the mixture (the list composes with the
    pass's own gather list); two runs bit-identical; more flagged workgroups than half the grid: the whole pass again."""
    ctx.set_heavy_rows(0)  # (round 6: by default such rows go round the fixed-point form and the second stage has nothing to do)
    n, d = 20000, 256
    rng = np.random.default_rng(50 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 17 + k)
    c, mu, s = 0.5 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.7
    m = P.PPCAModel(s, c, mu)
    w = rng.uniform(0.25, 2.0, n)
    ctx.set_grid_limit(8)
    per_wg = -(-(-(-n // 32)) // 8) * 32  # ceil(ceil(n / 32) / 8) tiles of 32 rows
    xo = x.copy()
    for r in (100, per_wg * 5 + 7, per_wg * 5 + 900):
        xo[r] *= 1e6
    for ww in (None, w):
        ds = P.Dataset(xo, ww)
        got = _stats(P, ds, m)
        assert ctx.last_guard() == (0, 1)
        mode, wgs, rows, _ = ctx.last_fallback()
        assert (mode, wgs, rows) == (2, 2, 2 * per_wg), (mode, wgs, rows)
        _assert_stats(got, oracle.stats(xo, s, c, mu, ww), d, k, 1e-9, (k, ww is None))
        assert np.array_equal(got, _stats(P, ds, m))  # (fixed summation orders in both stages)
        # the same rows without the outliers: nothing to do
        _stats(P, P.Dataset(x, ww), m)
        assert ctx.last_guard() == (0, 0) and ctx.last_fallback()[:3] == (0, 0, 0)
    # gathered: a third of the rows dropped by their weight, the outliers among the kept ones
    u = rng.uniform(-3.0, 0.0, n)
    u[rng.random(n) < 0.33] = -1e4
    u[[100, per_wg * 5 + 7, per_wg * 5 + 900]] = 0.0
    got, _, used = _gathered_stats(P, ctx, P.Dataset(xo), m, u)
    mode, wgs, rows, _ = ctx.last_fallback()
    assert mode == 2 and 1 <= wgs <= 3 and used < n, (mode, wgs, used)
    keep = u > -1e3
    wk = np.exp(u[keep] - u.max())
    want = oracle.stats(xo[keep], s, c, mu, wk)
    for name, a, b in _blocks(d, k)[:-1]:
        assert _rel(got[a:b], want[a:b]) < 1e-9, (name, k, "gathered")
    assert _rel(got[[-8, -7, -5]], want[[-8, -7, -5]]) < 1e-9  # (a component pass skips the log-likelihood: PassArgs::no_llk)
    # five of eight workgroups: more than half the grid -> the whole pass on the fp64 engine
    xa = x.copy()
    for g in range(5):
        xa[per_wg * g + 11] *= 1e6
    got = _stats(P, P.Dataset(xa), m)
    assert ctx.last_fallback()[0] == 1
    _assert_stats(got, oracle.stats(xa, s, c, mu), d, k, 1e-9, (k, "half the grid"))


@pytest.mark.parametrize("k,d", [(10, 256), (4, 64), (7, 200)])
def test_outlier_rows_go_round_the_fixed_point_form(P, oracle, ctx, k, d):
    """Round 6 (em9_kernel's back role, cold path): the rows of a tile that do not fit the fixed-point form of the mask-side statistics
    -- samples at 1e6 x their neighbours, a sample weight of 1e9 -- are cut out of the tile's digit planes and added to the fp64
    accumulators directly, under the old exponents: what the reference's f64 sums do with such a row (ppca_model.rs:297-306).  No
    exponent rises (counter 0 stays 0), no guard trips, nothing is recomputed; every block within 1e-9 of the oracle and, dimension by
    dimension where a large row is masked, the diagonal of S too -- on the full grid and on ONE workgroup walking every tile
    (rounds 3-5 had that workgroup cut every later row at the outlier's scale).  Up to eight such rows in one tile go round; nine are
    a change of scale: the exponents rise and the guard decides as before."""
    n = 6000
    rng = np.random.default_rng(31 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 19)
    c, mu, s = 0.5 * rng.standard_normal((d, k)), 0.05 * rng.standard_normal(d), 0.7
    m = P.PPCAModel(s, c, mu)
    kp = k * (k + 1) // 2
    diag = [a * (a + 1) // 2 + a for a in range(k)]

    def check(xo, w, rows, tag, expect_round):
        want = oracle.stats(xo, s, c, mu, w)
        for cap in (0, 1):
            ctx.set_grid_limit(cap)
            ctx.debug_counters(reset=True)
            got = _stats(P, P.Dataset(xo, w) if w is not None else P.Dataset(xo), m)
            cnt = ctx.debug_counters()
            if expect_round:
                assert ctx.last_guard() == (0, 0) and ctx.last_fallback()[:3] == (0, 0, 0), (tag, cap, ctx.last_guard())
                assert cnt[0] == 0, (tag, cap, cnt)
            _assert_stats(got, want, d, k, 1e-9, (k, cap, tag))
            Sg, Sw = got[d * k:d * k + d * kp].reshape(d, kp), want[d * k:d * k + d * kp].reshape(d, kp)
            for r in rows:
                masked = ~np.isfinite(xo[r])
                if masked.any():
                    assert (np.abs(Sg[masked][:, diag] - Sw[masked][:, diag]) / np.abs(Sw[masked][:, diag])).max() < 1e-9, (tag, cap, r)
        ctx.set_grid_limit(0)

    check(x, None, [], "clean", True)
    xo = x.copy()
    xo[100] *= 1e6
    check(xo, None, [100], "one row", True)
    xo = x.copy()
    rows = [64, 65, 70, 77, 80, 90, 94, 95, 3000, 5999]  # eight in one tile, the last row of the dataset
    for r in rows:
        xo[r] *= 10.0 ** rng.uniform(4, 8)
    check(xo, None, rows, "eight in a tile", True)
    w = rng.uniform(0.5, 2.0, n)
    w[1234] = 1e9
    w[17] = 1e-9
    check(x, w, [1234], "heavy weight", True)
    xo = x.copy()
    for r in range(128, 137):  # nine rows of one tile: a change of scale -- exponents rise, the guard behind them decides
        xo[r] *= 1e6
    check(xo, None, list(range(128, 137)), "nine in a tile", False)
    # an EM step on the outlier data agrees with the oracle's noise level (the row systems there have condition ~1e12)
    xo = x.copy()
    xo[100] *= 1e6
    s1, _, _ = oracle.iterate(xo, s, c, mu)
    new = m.iterate(P.Dataset(xo))
    assert abs(new.isotropic_noise - s1) < 1e-9 * s1 and np.isfinite(new.transform).all()


def test_slice_table_is_cached_per_model_content(P, oracle, ctx):
    """A fused pass builds the int8 slice table / guard flags of its model (qprep_kernel) unless the context's table is already
    that model's: the EM step's finalisation builds the NEW model's in the same launch (finalize_qprep_kernel), and a pass of the
    model the previous pass used skips it too.  The cache key is (buffer, write stamp): alternating models, a model whose buffer
    is re-used by a later step, and the guard's verdict for a model the cache served must all behave as with PPCA_QPREP_CACHE=0
    (every pass builds its table) -- bit for bit."""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from oracle import ppca_oracle as o
o.build()
x, _, _ = o.synth(3000, 200, 7, 0.3, 5)
rng = np.random.default_rng(1)
ds = P.Dataset(x)
A = P.PPCAModel(0.8, 0.5 * rng.standard_normal((200, 7)), np.zeros(200))
B = P.PPCAModel(0.3, rng.standard_normal((200, 7)), 0.1 * rng.standard_normal(200))
cb = rng.standard_normal((200, 7)); cb[:100] *= 1e-4
Bad = P.PPCAModel(1e-5, cb, np.zeros(200))  # (trips the Gram guard)
out = [A.llk(ds), B.llk(ds), A.llk(ds), Bad.llk(ds), A.llk(ds)]
m = A
for _ in range(4):
    m, l = m.iterate_with_llk(ds)
    out += [l, m.isotropic_noise, float(np.abs(m.transform).sum())]
out += [m.llk(ds), B.llk(ds)]
m2, l2 = Bad.iterate_with_llk(ds)
out += [l2, m2.isotropic_noise, _lib.default_context().last_guard()[0]]
m3, l3 = m.iterate_with_llk(ds)
out += [l3, m3.isotropic_noise, _lib.default_context().last_guard()[0]]
np.save(sys.argv[1], np.array(out, dtype=np.float64))
""" % root
    res = []
    with tempfile.TemporaryDirectory() as td:
        for env in ({}, {"PPCA_QPREP_CACHE": "0"}):
            out = os.path.join(td, "o%d.npy" % len(res))
            subprocess.run([sys.executable, "-c", code, out], check=True, env={**os.environ, **env}, timeout=600)
            res.append(np.load(out))
    assert np.array_equal(res[0], res[1]), (res[0], res[1])
    assert res[0][-1] == 0 and res[0][-4] == 1  # (the cached table's flags are the model's own: Bad trips the guard, its successor's model does not)


@pytest.mark.parametrize("k,d", [(4, 300), (20, 70)])
def test_outlier_row_on_the_split_pipeline(P, oracle, ctx, k, d):
    """The same outlier row on shapes of the generic split pipeline (its int8 statistics product has a per-chunk guard of its
    own -- column maximum against mean magnitude -- with the fp64 GEMM behind it): every block within 1e-9 of the oracle."""
    n = 3000
    rng = np.random.default_rng(5)
    x, _, _ = oracle.synth(n, d, k, 0.3, 13)
    x[100] *= 1e6
    c, mu, s = 0.5 * rng.standard_normal((d, k)), np.zeros(d), 0.7
    m = P.PPCAModel(s, c, mu)
    got, want = _stats(P, P.Dataset(x), m), oracle.stats(x, s, c, mu)
    _assert_stats(got, want, d, k, 1e-9, (k, d))
    kp = k * (k + 1) // 2
    diag = [a * (a + 1) // 2 + a for a in range(k)]
    Sg, Sw = got[d * k:d * k + d * kp].reshape(d, kp), want[d * k:d * k + d * kp].reshape(d, kp)
    masked = ~np.isfinite(x[100])
    assert (np.abs(Sg[masked][:, diag] - Sw[masked][:, diag]) / np.abs(Sw[masked][:, diag])).max() < 1e-9


def test_heavy_tailed_fuzz_of_the_fixed_point_statistics():
    """tools/fuzz_gpu2.py: 24 random cases over the three engines (eight-wave kernel, two-kernel pass, split pipeline) with
    outlier rows (1e2 .. 1e8 x), heavy-tailed weights (log-normal, sigma up to 12), capped grids: the statistics block by block
    (1e-8) and the diagonal of S and the totals DIMENSION BY DIMENSION (1e-7 element-wise) against the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu2.py"), "7", "24"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz2 ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


@pytest.mark.parametrize("em9", ["0", "1"])
def test_both_eight_wave_kernels_against_the_oracle(em9):
    """tools/em9_check.py in a child process (the switch is read once per process): em9_kernel (the default: the solve
    pipelined across tiles) and em8_kernel (PPCA_EM9=0, round 3's) at N = 20 000 on the full grid and on 2 / 1 workgroups,
    weighted and not, and on ragged shapes down to one row -- every block of the statistics within 1e-9 of the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PPCA_EM9=em9)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "em9_check.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "em9 check ok" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
