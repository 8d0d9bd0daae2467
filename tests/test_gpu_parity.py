"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the
committed golden vectors.  Tolerance: 1e-5 relative (BASELINE.json north_star, fp64);
most checks are held far tighter because both sides are fp64."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
              if not os.path.basename(p).startswith(("mix_", "cfg5_")))
GOLD_MIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mix_*.npz")))
RTOL = 1e-5  # north_star tolerance


@pytest.fixture(scope="module")
def P(hiplib):
    import ppca_rs_amd as p

    return p


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_mfma_f64_lane_map(P):
    """The C/D lane map the kernels assume for v_mfma_f64_16x16x4_f64 (asymmetric data)."""
    from ppca_rs_amd import _lib

    rng = np.random.default_rng(0)
    a = rng.integers(-5, 6, (16, 4)).astype(np.float64)
    b = rng.integers(-5, 6, (4, 16)).astype(np.float64)
    out = np.zeros((16, 16))
    ctx = _lib.default_context()
    _lib.check(_lib.lib().ppca_debug_mfma_probe(ctx.handle, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out)))
    np.testing.assert_array_equal(out, a @ b)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_golden(P, path):
    g = np.load(path)
    x, s, c, mu = g["x"], float(g["s0"]), g["c0"], g["mu0"]
    w = g["w"] if "w" in g else None
    ds = P.Dataset(x, w)
    m = P.PPCAModel(s, c, mu)
    assert _rel(m.llks(ds), g["llks"]) < 1e-9
    assert abs(m.llk(ds) - g["llk"]) < 1e-9 * abs(g["llk"])
    inf = m.infer(ds)
    assert _rel(inf.states(), g["states"]) < 1e-8
    assert _rel(np.array(inf.covariances()), g["covs"]) < 1e-8
    assert _rel(m.smooth(ds).numpy(), g["smooth"]) < 1e-9
    ex = m.extrapolate(ds).numpy()
    assert _rel(ex, g["extrapolate"]) < 1e-9
    ob = np.isfinite(x)
    np.testing.assert_array_equal(ex[ob], x[ob])  # observed values pass through bit-exactly
    if "prior" in os.path.basename(path):
        prior = (P.Prior().with_mean_prior(np.linspace(-1, 1, 8), 0.5 * np.eye(8) + 0.1)
                 .with_isotropic_noise_prior(3.0, 2.0).with_transformation_precision(0.7))
    else:
        prior = None
    for it in range(len(g["it_sigma"])):
        m, llk = m.iterate_with_llk(ds, prior)
        assert abs(llk - g["it_llk"][it]) < 1e-8 * abs(g["it_llk"][it])
        assert abs(m.isotropic_noise - g["it_sigma"][it]) < RTOL * 1e-2 * g["it_sigma"][it]
        assert _rel(m.transform, g["it_c"][it]) < RTOL * 1e-1
        assert _rel(m.mean, g["it_mean"][it]) < RTOL * 1e-1
    assert _rel(m.to_canonical().transform, g["canonical"]) < RTOL


@pytest.mark.parametrize("path", GOLD_MIX, ids=[os.path.basename(p) for p in GOLD_MIX])
def test_golden_mixture(P, path):
    """The HIP path against the committed mixture fixture (no oracle at run time): llks, inference outputs, three
    weighted mixture EM iterations."""
    g = np.load(path)
    x, w, sig, cs, ms, lw = g["x"], g["w"], g["sig0"], g["cs0"], g["ms0"], g["lw0"]
    ds = P.Dataset(x, w)
    mix = P.PPCAMix([P.PPCAModel(sig[c], cs[c], ms[c]) for c in range(len(sig))], lw)
    assert _rel(mix.llks(ds), g["llks"]) < 1e-9
    inf = mix.infer(ds)
    assert _rel(inf.log_posteriors(), g["inf_log_posterior"]) < 1e-8
    assert _rel(inf.states(), g["inf_state"]) < 1e-8 and _rel(np.array(inf.covariances()), g["inf_covariance"]) < 1e-8
    assert _rel(mix.smooth(ds).numpy(), g["inf_smooth"]) < 1e-8
    assert _rel(mix.extrapolate(ds).numpy(), g["inf_extrapolate"]) < 1e-8
    assert _rel(mix._mix_recon(ds, 2).numpy(), g["inf_smooth_covariance_diagonal"]) < 1e-8
    assert _rel(mix._mix_recon(ds, 3).numpy(), g["inf_extrapolate_covariance_diagonal"]) < 1e-8
    for it in range(len(g["it_llk"])):
        mix, llk = mix.iterate_with_llk(ds)
        assert abs(llk - g["it_llk"][it]) < 1e-8 * abs(g["it_llk"][it])
        for c, mdl in enumerate(mix.models):
            assert abs(mdl.isotropic_noise - g["it_sigma"][it][c]) < RTOL * 1e-1 * g["it_sigma"][it][c]
            assert _rel(mdl.transform, g["it_c"][it][c]) < RTOL and _rel(mdl.mean, g["it_mean"][it][c]) < RTOL
        assert _rel(mix.log_weights, g["it_lw"][it]) < RTOL


def test_stats_raw_against_oracle(P, oracle):
    from ppca_rs_amd import _lib

    x, _, _ = oracle.synth(1000, 256, 10, 0.3, 77)
    x[10] = np.nan
    rng = np.random.default_rng(5)
    c, mu, s = rng.standard_normal((256, 10)), 0.1 * rng.standard_normal(256), 0.7
    w = rng.uniform(0.5, 1.5, 1000)
    ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
    n = _lib.lib().ppca_stats_len(256, 10)
    got = np.empty(n)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    want = oracle.stats(x, s, c, mu, w)
    d, k, kp = 256, 10, 55
    bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, n]
    for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
        assert _rel(got[a:b], want[a:b]) < 1e-9, name


@pytest.mark.parametrize("n,d,k,mp", [(10_000, 32, 4, 0.0), (20_000, 256, 10, 0.3), (3000, 200, 7, 0.5), (33, 5, 1, 0.2)])
def test_em_iterations_match_oracle(P, oracle, n, d, k, mp):
    """BASELINE configs 1/2 at oracle-sized N: 1 and several iterations from a fixed start."""
    x, _, _ = oracle.synth(n, d, k, mp, 1000 + d)
    rng = np.random.default_rng(2000 + d)
    c, mu, s = rng.standard_normal((d, k)), np.zeros(d), 1.0
    ds, m = P.Dataset(x), P.PPCAModel(s, c, mu)
    iters = 10 if n <= 10_000 else 3
    for it in range(iters):
        want_llk = oracle.llk(x, s, c, mu)
        s, c, mu = oracle.iterate(x, s, c, mu)
        m, llk = m.iterate_with_llk(ds)
        assert abs(llk - want_llk) < RTOL * 1e-3 * abs(want_llk), it
        assert abs(m.isotropic_noise - s) < RTOL * s, it
        assert _rel(m.transform, c) < RTOL, it
        assert _rel(m.mean, mu) < RTOL, it
    assert _rel(m.to_canonical().transform, oracle.to_canonical(c)) < RTOL
    assert _rel(m.extrapolate(ds).numpy(), oracle.reconstruct(x, s, c, mu, "extrapolate")) < RTOL


def test_edge_cases(P, oracle):
    rng = np.random.default_rng(3)
    d, k = 16, 3
    c, mu, s = rng.standard_normal((d, k)), rng.standard_normal(d), 0.5
    m = P.PPCAModel(s, c, mu)
    # all samples fully masked -> llk 0, posterior N(0, I)
    x = np.full((5, d), np.nan)
    ds = P.Dataset(x)
    assert m.llk(ds) == 0.0
    inf = m.infer(ds)
    np.testing.assert_array_equal(inf.states(), np.zeros((5, k)))
    np.testing.assert_allclose(np.array(inf.covariances()), np.tile(np.eye(k), (5, 1, 1)), atol=1e-15)
    np.testing.assert_allclose(m.smooth(ds).numpy(), np.tile(mu, (5, 1)))
    assert ds.empty_dimensions() == list(range(d))
    # +-inf is masked and comes back NaN (dataset.rs:19-22, :64-72)
    x = rng.standard_normal((4, d))
    x[0, 1], x[1, 2], x[2, 3] = np.inf, -np.inf, np.nan
    ds = P.Dataset(x)
    back = ds.numpy()
    assert np.isnan(back[0, 1]) and np.isnan(back[1, 2]) and np.isnan(back[2, 3])
    xn = np.where(np.isfinite(x), x, np.nan)
    assert _rel(m.llks(ds), oracle.llks(xn, s, c, mu)) < 1e-10
    # ragged: N not a multiple of the 32-sample tile, single row, strided / transposed views
    for n in (1, 31, 32, 33, 65):
        x = rng.standard_normal((n, d))
        x[rng.random((n, d)) < 0.4] = np.nan
        assert _rel(m.llks(P.Dataset(x)), oracle.llks(x, s, c, mu)) < 1e-10
    big = rng.standard_normal((d, 40))
    view = big.T[::2]  # non-contiguous (20, d)
    assert _rel(P.Dataset(view).numpy(), np.ascontiguousarray(view)) == 0.0
    # an empty dimension keeps its row and mean (ppca_model.rs:313-321, :373-377)
    x = rng.standard_normal((200, d))
    x[:, 7] = np.nan
    ds = P.Dataset(x)
    assert ds.empty_dimensions() == [7]
    new = m.iterate(ds)
    np.testing.assert_array_equal(new.transform[7], c[7])
    assert new.mean[7] == mu[7]
    s1, c1, m1 = oracle.iterate(x, s, c, mu)
    assert _rel(new.transform, c1) < 1e-8 and abs(new.isotropic_noise - s1) < 1e-10
    # shape mismatch and empty dataset raise
    with pytest.raises(P.PPCAError):
        m.llk(P.Dataset(np.zeros((3, d + 1))))
    with pytest.raises(ValueError):
        m.iterate(P.Dataset(np.zeros((0, d))))


def test_dataset_ops(P):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((10, 4))
    x[2, 1] = np.nan
    w = np.arange(10, dtype=np.float64) + 1
    ds = P.Dataset(x, w)
    assert len(ds) == 10 and ds.output_size() == 4
    np.testing.assert_array_equal(ds.weights(), w)
    chunks = list(ds.chunks(3))  # stride ceil(10/3) = 4
    assert [len(c) for c in chunks] == [4, 4, 2]
    np.testing.assert_array_equal(chunks[1].weights(), w[4:8])
    cat = P.Dataset.concat(chunks)
    np.testing.assert_array_equal(np.nan_to_num(cat.numpy(), nan=-1), np.nan_to_num(x, nan=-1))
    np.testing.assert_array_equal(cat.weights(), w)
    re = ds.with_weights(np.ones(10))
    np.testing.assert_array_equal(re.weights(), np.ones(10))
    np.testing.assert_array_equal(np.isnan(re.numpy()), np.isnan(x))
    ds2 = P.Dataset.load(ds.dump())
    np.testing.assert_array_equal(ds2.weights(), w)


def test_trainer_and_sampling(P, capsys):
    """examples/toy_model.py flow: sample -> init -> train -> canonical, llk never decreases."""
    real = P.PPCAModel(0.1, np.array([[1, 1], [0, 1], [0, 1]], dtype="float64"), np.array([[0], [1], [0]], dtype="float64"))
    sample = real.sample(2000, 0.2, seed=7)
    assert len(sample) == 2000
    frac = np.isnan(sample.numpy()).mean()
    assert 0.15 < frac < 0.25
    np.testing.assert_array_equal(np.nan_to_num(sample.numpy()), np.nan_to_num(real.sample(2000, 0.2, seed=7).numpy()))
    model = P.PPCATrainer(sample).train(state_size=2, n_iters=30, metric="llk", seed=3)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Masked PPCA iteration")]
    assert len(lines) == 30 and lines[0].startswith("Masked PPCA iteration 1: llk=")
    vals = [float(l.split("=")[1]) for l in lines]
    assert all(b >= a - 1e-9 for a, b in zip(vals, vals[1:]))
    assert model.isotropic_noise < 0.2
    inf = model.infer(sample)
    diag = inf.smoothed_covariances_diagonal(model).numpy()
    assert diag.shape == (2000, 3) and (diag > 0).all()


def test_mixture_against_oracle(P, oracle):
    rng = np.random.default_rng(9)
    d, k, nm, n = 24, 3, 3, 1500
    parts = []
    for c_ in range(nm):
        xx, _, _ = oracle.synth(n // nm, d, k, 0.25, 300 + c_, mean_scale=3.0)
        parts.append(xx)
    x = np.concatenate(parts)
    sig = np.array([1.0, 1.2, 0.9])
    cs = rng.standard_normal((nm, d, k))
    ms = rng.standard_normal((nm, d))
    lw = np.log(np.array([0.3, 0.3, 0.4]))
    ds = P.Dataset(x)
    mix = P.PPCAMix([P.PPCAModel(sig[c_], cs[c_], ms[c_]) for c_ in range(nm)], lw)
    assert _rel(mix.llks(ds), oracle.mix_llks(x, sig, cs, ms, lw)) < 1e-9
    assert _rel(mix.infer_cluster(ds), oracle.mix_infer_cluster(x, sig, cs, ms, lw)) < 1e-8
    for _ in range(3):
        want = oracle.mix_iterate(x, sig, cs, ms, lw)
        mix, llk = mix.iterate_with_llk(ds)
        assert abs(llk - oracle.mix_llks(x, sig, cs, ms, lw).sum()) < 1e-8 * abs(llk)
        sig, cs, ms, lw = want
        for c_, mdl in enumerate(mix.models):
            assert abs(mdl.isotropic_noise - sig[c_]) < RTOL * sig[c_]
            assert _rel(mdl.transform, cs[c_]) < RTOL and _rel(mdl.mean, ms[c_]) < RTOL
        assert _rel(mix.log_weights, lw) < RTOL


def test_mixture_of_three_hundred_components(P, oracle):
    """mix.rs:50-71 puts no limit on the number of components: the step's small device vectors are sized by it
    (rounds 1-4 refused more than 256).  Broad components, so that every one keeps weight on every sample."""
    rng = np.random.default_rng(5)
    d, k, nm, n = 8, 2, 300, 400
    x = rng.standard_normal((n, d))
    x[rng.random((n, d)) < 0.1] = np.nan
    sig = rng.uniform(0.8, 1.2, nm)
    cs = 0.3 * rng.standard_normal((nm, d, k))
    ms = 0.3 * rng.standard_normal((nm, d))
    lw = np.log(rng.dirichlet(np.ones(nm) * 5))
    ds = P.Dataset(x)
    mix = P.PPCAMix([P.PPCAModel(sig[c_], cs[c_], ms[c_]) for c_ in range(nm)], lw)
    # (thousands of oracle calls over 400 samples each: on a 128-core host the OpenMP teams cost 20 minutes; two threads: seconds)
    old_threads = oracle.set_threads(2)
    try:
        assert _rel(mix.llks(ds), oracle.mix_llks(x, sig, cs, ms, lw)) < 1e-9
        assert _rel(mix.infer_cluster(ds), oracle.mix_infer_cluster(x, sig, cs, ms, lw)) < 1e-8
        for _ in range(2):
            want = oracle.mix_iterate(x, sig, cs, ms, lw)
            mix, llk = mix.iterate_with_llk(ds)
            assert abs(llk - oracle.mix_llks(x, sig, cs, ms, lw).sum()) < 1e-8 * abs(llk)
            sig, cs, ms, lw = want
            for c_, mdl in enumerate(mix.models):
                assert abs(mdl.isotropic_noise - sig[c_]) < RTOL * sig[c_]
                assert _rel(mdl.transform, cs[c_]) < RTOL and _rel(mdl.mean, ms[c_]) < RTOL
            assert _rel(mix.log_weights, lw) < RTOL
        assert _rel(mix.smooth(ds).numpy(), oracle.mix_inferred(x, sig, cs, ms, lw)["smooth"]) < 1e-8
    finally:
        oracle.set_threads(old_threads)


def test_mixture_inference_outputs_against_oracle(P, oracle):
    """PPCAMix.infer / smooth / extrapolate and the InferredMaskedMix accessors (mix.rs:179-265,
    :352-515) -- GPU per-component inference + posterior-weighted combination vs the oracle's restatement."""
    rng = np.random.default_rng(19)
    d, k, nm, n = 20, 3, 3, 600
    x = np.concatenate([oracle.synth(n // nm, d, k, 0.3, 500 + c_, mean_scale=2.0)[0] for c_ in range(nm)])
    x[7] = np.nan  # an all-masked sample
    sig = np.array([0.8, 1.1, 0.6])
    cs = rng.standard_normal((nm, d, k))
    ms = rng.standard_normal((nm, d))
    lw = np.log(np.array([0.5, 0.2, 0.3]))
    want = oracle.mix_inferred(x, sig, cs, ms, lw)
    w = rng.uniform(0.5, 2.0, x.shape[0])
    ds = P.Dataset(x, w)
    mix = P.PPCAMix([P.PPCAModel(sig[c_], cs[c_], ms[c_]) for c_ in range(nm)], lw)
    sm, ex = mix.smooth(ds), mix.extrapolate(ds)
    assert _rel(sm.numpy(), want["smooth"]) < 1e-8 and _rel(ex.numpy(), want["extrapolate"]) < 1e-8
    ob = np.isfinite(x)
    assert np.abs(ex.numpy()[ob] - x[ob]).max() < 1e-12 * np.abs(x[ob]).max()  # posteriors sum to one
    assert np.all(sm.weights() == 1.0)  # the reference's mixture outputs drop the weights (mix.rs:245-265)
    assert np.all(np.isfinite(sm.numpy()))
    inf = mix.infer(ds)
    assert _rel(inf.log_posteriors(), want["log_posterior"]) < 1e-8
    assert _rel(inf.posteriors().sum(axis=1), np.ones(x.shape[0])) < 1e-12
    assert _rel(inf.states(), want["state"]) < 1e-8
    assert _rel(np.array(inf.covariances()), want["covariance"]) < 1e-8
    assert _rel(inf.smoothed(mix).numpy(), want["smooth"]) < 1e-8
    assert _rel(inf.extrapolated(mix, ds).numpy(), want["extrapolate"]) < 1e-8
    assert _rel(inf.smoothed_covariances_diagonal(mix).numpy(), want["smooth_covariance_diagonal"]) < 1e-8
    assert _rel(inf.extrapolated_covariances_diagonal(mix, ds).numpy(), want["extrapolate_covariance_diagonal"]) < 1e-8
    # device-side diagonals (ppca_mix_reconstruct modes 2 and 3) agree with the accessor path
    assert _rel(mix._mix_recon(ds, 2).numpy(), want["smooth_covariance_diagonal"]) < 1e-8
    assert _rel(mix._mix_recon(ds, 3).numpy(), want["extrapolate_covariance_diagonal"]) < 1e-8
    # full covariances: diagonal of the d x d matrices = the diagonal accessor; extrapolated uses the smoothed
    # component covariances, as written upstream (mix.rs:464-477)
    full = np.array(inf.smoothed_covariances(mix)[:5])
    assert _rel(np.einsum("njj->nj", full), want["smooth_covariance_diagonal"][:5]) < 1e-8
    assert len(inf.extrapolated_covariances(mix, ds)) == x.shape[0]
    # samplers: shapes, determinism under a seed, round trip of the container
    s1, s2 = inf.posterior_sampler().sample(seed=4).numpy(), inf.posterior_sampler().sample(seed=4).numpy()
    assert s1.shape == x.shape and np.array_equal(s1, s2)
    gen = mix.sample(500, 0.2, seed=5).numpy()
    assert gen.shape == (500, d) and 0.1 < np.isnan(gen).mean() < 0.3
    back = P.PPCAMix.load(mix.dump())
    assert _rel(back.llks(ds), mix.llks(ds)) < 1e-14
    import pickle
    assert _rel(pickle.loads(pickle.dumps(mix)).log_weights, mix.log_weights) < 1e-15


def test_compiled_host_example_runs():
    """The C-ABI driven from a compiled host (examples/em_train.cpp): EM on the fused path, llk monotone, the
    noise level of the generating model recovered."""
    import subprocess

    from ppca_rs_amd import build

    exe = build.build_example()
    out = subprocess.run([exe, "200000", "256", "10", "12"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("Masked PPCA iteration")]
    llks = [float(l.split("llk=")[1]) for l in lines]
    assert len(llks) == 12 and all(b >= a for a, b in zip(llks, llks[1:]))
    sigma = float(out.stdout.split("fitted isotropic noise ")[1].split()[0])
    assert 0.05 < sigma < 0.5


def test_dataframe_adapter_round_trip(P):
    """DataFrame -> Dataset -> model -> DataFrame (python/ppca_rs/__init__.py:119-433)."""
    import pandas as pd

    rng = np.random.default_rng(8)
    n_s, n_d = 300, 6
    full = rng.standard_normal((n_s, 2)) @ rng.standard_normal((2, n_d)) + 0.05 * rng.standard_normal((n_s, n_d))
    keep = rng.random((n_s, n_d)) < 0.8
    si, di = np.nonzero(keep)
    df = pd.DataFrame({"unit": si, "sensor": [f"s{j}" for j in di], "value": full[si, di]}).sample(frac=1.0, random_state=1)
    ad = P.DataFrameAdapter.from_pandas(df, keys=["unit"], dimensions=["sensor"], metric="value")
    x = ad.dataset.numpy()
    assert x.shape == (n_s, n_d) and np.array_equal(np.isfinite(x), keep) and np.allclose(x[keep], full[keep])
    model = P.PPCATrainer(ad.dataset).train(state_size=2, n_iters=15, quiet=True)
    long = ad.convert_datasets({"seen": ad.dataset, "filled": model.extrapolate(ad.dataset)})
    assert list(long.columns) == ["unit", "sensor", "seen", "filled"] and len(long) == n_s * n_d
    merged = long.merge(df, on=["unit", "sensor"])
    assert np.allclose(merged["filled"], merged["value"]) and np.allclose(merged["seen"], merged["value"])
    assert long["filled"].notna().all() and long["seen"].isna().sum() == (~keep).sum()
    again = ad.description().adapt_pandas(df)
    assert np.array_equal(np.nan_to_num(again.dataset.numpy()), np.nan_to_num(x))


def test_dump_load_both_containers(P):
    """dump()/load() (src/python_bindings.rs:66-79, :388-401): own npz container and the restated bincode layout."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal((50, 33))
    x[rng.random(x.shape) < 0.25] = np.nan
    ds = P.Dataset(x, rng.uniform(0.5, 1.5, 50))
    m = P.PPCAModel(0.3, rng.standard_normal((33, 4)), rng.standard_normal(33))
    for fmt in ("npz", "bincode"):
        d2, m2 = P.Dataset.load(ds.dump(fmt)), P.PPCAModel.load(m.dump(fmt))
        assert np.array_equal(np.nan_to_num(d2.numpy()), np.nan_to_num(x)) and np.array_equal(d2.weights(), ds.weights())
        assert m2.isotropic_noise == 0.3 and np.array_equal(m2.transform, m.transform) and np.array_equal(m2.mean, m.mean)
        assert m2.llk(d2) == m.llk(ds)
    with pytest.raises(Exception):
        P.PPCAModel.load(m.dump("bincode")[:-5])


@pytest.mark.parametrize("k", list(range(1, 11)))
def test_fused_path_shape_sweep(P, oracle, k):
    """Every state size of the fused kernel (each is its own instantiation: different tile counts, piece
    schedules, 4x4x4 split only at k = 10) x ragged d and N, weights, all-masked rows: statistics buffer, llks,
    posteriors and all four output passes against the oracle."""
    from ppca_rs_amd import _lib

    rng = np.random.default_rng(40 + k)
    for d, n in ((256, 97), (255, 64), (200, 33), (64, 129), (max(k, 3), 31)):
        assert _lib.lib().ppca_path_kind(d, k) == 1
        x, _, _ = oracle.synth(n, d, k, 0.35, 900 + 10 * k + d)
        x[n // 2] = np.nan
        w = rng.uniform(0.25, 2.0, n)
        c, mu, s = 0.5 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), 0.6
        ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
        L = _lib.lib().ppca_stats_len(d, k)
        got = np.empty(L)
        _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
        want = oracle.stats(x, s, c, mu, w)
        kp = k * (k + 1) // 2
        bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
        for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
            assert _rel(got[a:b], want[a:b]) < 1e-9, (name, d, n)
        # the same, unweighted (the other llk path of the EM pass: one logarithm per kernel)
        dsu = P.Dataset(x)
        _lib.check(_lib.lib().ppca_stats_raw(dsu._ctx.handle, dsu._h, m._device(dsu._ctx).h, _lib.ptr(got)))
        want = oracle.stats(x, s, c, mu)
        assert _rel(got[bounds[-2]:], want[bounds[-2]:]) < 1e-9 and _rel(got[:bounds[-2]], want[:bounds[-2]]) < 1e-9
        assert _rel(m.llks(ds), oracle.llks(x, s, c, mu)) < 1e-10
        st, cv = oracle.infer(x, s, c, mu)
        inf = m.infer(ds)
        assert _rel(inf.states(), st) < 1e-8 and _rel(np.array(inf.covariances()), cv) < 1e-8
        assert _rel(m.smooth(ds).numpy(), oracle.reconstruct(x, s, c, mu, "smooth")) < 1e-9
        ex = m.extrapolate(ds).numpy()
        assert _rel(ex, oracle.reconstruct(x, s, c, mu, "extrapolate")) < 1e-9
        assert np.array_equal(ex[np.isfinite(x)], x[np.isfinite(x)])
        assert _rel(inf.smoothed_covariances_diagonal(m).numpy(), oracle.covariance_diagonal(x, s, c, mu, "smooth")) < 1e-8
        for mode, name in ((0, "smooth"), (1, "extrapolate")):
            h = C.c_void_p()
            _lib.check(_lib.lib().ppca_covariance_diagonal(ds._ctx.handle, ds._h, m._device(ds._ctx).h, mode, C.byref(h)))
            assert _rel(P.Dataset._wrap(h, ds._ctx).numpy(), oracle.covariance_diagonal(x, s, c, mu, name)) < 1e-8


@pytest.mark.parametrize("k", list(range(11, 17)))
def test_two_kernel_em_pass_shape_sweep(P, oracle, k):
    """State sizes 11..16 at d <= 256 (the reference's own largest workload is d = 200, k = 16: lib.rs:82-99) run the EM
    pass as the two fused kernels of ppca_em16.hip -- each k its own instantiation (k <= 13: factor in registers, k >= 14:
    split over lane pairs and parked in LDS): ragged d and N, weights and none, an all-masked row, more than one chunk
    of the hand-over buffer, statistics buffer against the oracle and two EM iterations end to end."""
    from ppca_rs_amd import _lib

    rng = np.random.default_rng(140 + k)
    for d, n in ((256, 97), (255, 64), (200, 33), (64, 129), (k + 3, 31), (200, 700)):
        assert _lib.lib().ppca_path_kind(d, k) == 0  # (the other passes of these shapes stay on the split pipeline)
        x, _, _ = oracle.synth(n, d, k, 0.35, 900 + 10 * k + d)
        x[n // 2] = np.nan
        w = rng.uniform(0.25, 2.0, n)
        c, mu, s = 0.5 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), 0.6
        m = P.PPCAModel(s, c, mu)
        L = _lib.lib().ppca_stats_len(d, k)
        got = np.empty(L)
        kp = k * (k + 1) // 2
        bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
        for weights in (w, None):
            ds = P.Dataset(x, weights)
            _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
            want = oracle.stats(x, s, c, mu, weights)
            for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
                assert _rel(got[a:b], want[a:b]) < 1e-9, (name, d, n, weights is None)
    ds = P.Dataset(x, w)
    for _ in range(2):
        want_llk = oracle.llk(x, s, c, mu, w)
        s, c, mu = oracle.iterate(x, s, c, mu, w)
        m, llk = m.iterate_with_llk(ds)
        assert abs(llk - want_llk) < 1e-9 * abs(want_llk)
        assert abs(m.isotropic_noise - s) < RTOL * s and _rel(m.transform, c) < RTOL and _rel(m.mean, mu) < RTOL


@pytest.mark.parametrize("k", [12, 16])
def test_two_kernel_em_pass_rescales_its_fixed_point_form(P, oracle, k):
    """sstat16_kernel cuts the rows [wP | wz | w] against per-column exponents set by the first tile; weights growing by
    2^80 over the rows force the cold path again and again (a tile that does not fit: flush the int64 accumulators into
    the partial, raise the exponents, cut again), descending weights leave the exponents where the first tile put
    them (precision is then relative to the LARGEST terms, as in any fp64 sum): both against the oracle."""
    from ppca_rs_amd import _lib

    d, n = 64, 4000
    rng = np.random.default_rng(7 + k)
    x, _, _ = oracle.synth(n, d, k, 0.3, 321 + k)
    c, mu, s = 0.4 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.8
    m = P.PPCAModel(s, c, mu)
    L = _lib.lib().ppca_stats_len(d, k)
    kp = k * (k + 1) // 2
    bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
    for w in (2.0 ** np.linspace(-40, 40, n), 2.0 ** np.linspace(40, -40, n)):
        ds = P.Dataset(x, w)
        got = np.empty(L)
        _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
        want = oracle.stats(x, s, c, mu, w)
        for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
            assert _rel(got[a:b], want[a:b]) < 1e-9, (name, k, w[0] < w[-1])


def test_two_kernel_em_pass_chunks_and_guard(P, oracle):
    """ppca_em16.hip across several chunks of its hand-over buffer (PPCA_GEN_CHUNK, read per call) and on a model that
    trips the int8 Gram guard (Gram rows from the fp64 product instead): both against the oracle; and the switch back to
    the split pipeline (PPCA_EM16=0 is read once per process: checked in a child)."""
    import subprocess, sys
    from ppca_rs_amd import _lib

    d, k, n = 200, 16, 700
    rng = np.random.default_rng(3)
    x, _, _ = oracle.synth(n, d, k, 0.3, 77)
    w = rng.uniform(0.5, 1.5, n)
    c, mu, s = 0.3 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.9
    L = _lib.lib().ppca_stats_len(d, k)
    want = oracle.stats(x, s, c, mu, w)
    code = ("import os, sys, numpy as np; sys.path.insert(0, %r); import ppca_rs_amd as P; from ppca_rs_amd import _lib;"
            "g = np.load(sys.argv[1]); ds, m = P.Dataset(g['x'], g['w']), P.PPCAModel(float(g['s']), g['c'], g['mu']);"
            "got = np.empty(int(g['L'])); _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)));"
            "np.save(sys.argv[2], got)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), x=x, w=w, s=s, c=c, mu=mu, L=L)
        for env in ({"PPCA_GEN_CHUNK": "256"}, {"PPCA_EM16": "0"}):
            out = os.path.join(td, "out.npy")
            subprocess.run([sys.executable, "-c", code, os.path.join(td, "in.npz"), out], check=True, env={**os.environ, **env}, timeout=600)
            assert _rel(np.load(out), want) < 1e-9, env
    # rows spanning 1e8 and a tiny sigma: both bounds of the guard fail -> fp64 Gram rows (ill-conditioned by construction)
    c2 = c.copy(); c2[: d // 2] *= 1e-4
    m2, ds = P.PPCAModel(1e-5, c2, mu), P.Dataset(x, w)
    eng = C.c_int32(-1)
    _lib.check(_lib.lib().ppca_gram_engine(ds._ctx.handle, m2._device(ds._ctx).h, C.byref(eng)))
    assert eng.value == 1
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m2._device(ds._ctx).h, _lib.ptr(got)))
    assert _rel(got, oracle.stats(x, 1e-5, c2, mu, w)) < 1e-4


def test_gram_guard_of_the_ninth_tile_at_k16(P, oracle):
    """k = 16 has NINE packed-column tiles; the ninth (pairs (15, 8..15)) files its verdict in qflag[8], the slot the fused
    k <= 10 pass uses as its run-again flag.  Round 4's qprep_kernel cleared that slot from block 0 in every instantiation,
    unordered against block 8's write: a model whose ONLY unsafe tile is the ninth could run on the int8 Gram.  Column 15 of
    C at 1e3 x the others (entries of equal magnitude), one weak row (so the smallest row norm is small) and a small sigma
    trip exactly that tile: the guard must report the fp64 engine every time."""
    from ppca_rs_amd import _lib

    d, k, n = 200, 16, 300
    rng = np.random.default_rng(16)
    x, _, _ = oracle.synth(n, d, k, 0.3, 1616)
    c, mu, s = 0.3 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.03
    c[:, 15] = 300.0 * np.where(rng.random(d) < 0.5, -1.0, 1.0)
    c[0] *= 0.1
    c[0, 15] = 0.0
    # the guard's own arithmetic (ppca_kernels.hip, qprep_kernel), restated: tile t is unsafe iff neither bound holds
    eps = lambda cmax: 2.0 ** (np.frexp(cmax)[1] - 63)
    rmin = (c * c).sum(1).min()
    unsafe = []
    for t in range(9):
        bad = False
        for cc in range(16 * t, min(16 * t + 16, k * (k + 1) // 2)):
            a = int((np.sqrt(8 * cc + 1) - 1) // 2)
            b = cc - a * (a + 1) // 2
            e = eps(np.abs(c[:, a] * c[:, b]).max())
            bad |= not (k * d * e <= 1e-8 * s * s or k * k * e <= 2.0 ** -40 * rmin)
        unsafe.append(bad)
    assert unsafe == [False] * 8 + [True], unsafe
    m, ds = P.PPCAModel(s, c, mu), P.Dataset(x)
    for _ in range(25):  # (the race needed block 0 to finish after block 8)
        eng = C.c_int32(-1)
        _lib.check(_lib.lib().ppca_gram_engine(ds._ctx.handle, m._device(ds._ctx).h, C.byref(eng)))
        assert eng.value == 1
    L = _lib.lib().ppca_stats_len(d, k)
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    # (|G| / sigma^2 ~ 2e10: the oracle's subtractive form keeps about six digits here)
    assert _rel(got, oracle.stats(x, s, c, mu, None)) < 1e-4


def test_reference_usage_patterns():
    """The call patterns of the reference's own examples (examples/*.py: keyword construction, np.matrix and
    transposed inputs, row or column means, positional mask probability, the `ppca_rs.ppca_rs` submodule, pickling,
    priors, the mixture trainer, a k = 16 model on the generic path) run unchanged against this package."""
    import pickle

    import ppca_rs
    from ppca_rs import Dataset, PPCAMix, PPCAMixTrainer, Prior
    from ppca_rs.ppca_rs import PPCAModel

    real = PPCAModel(transform=np.array([[1, 1], [0, 1], [0, 1]], dtype="float64"), isotropic_noise=0.1,
                     mean=np.array([[0], [1], [0]], dtype="float64"))
    sample = real.sample(100, mask_prob=0.2)
    model = PPCAModel.init(2, sample)
    llks = []
    for _ in range(30):
        llks.append(model.llk(sample))
        model = model.iterate(sample)
    assert all(b >= a - 1e-9 * abs(a) for a, b in zip(llks, llks[1:])) and len(sample) == 100
    model = model.to_canonical()
    assert "PPCAModel" in repr(model) and model.singular_values.shape == (2,)
    sd = model.infer(sample).smoothed_covariances_diagonal(model).numpy() ** 0.5
    assert sd.shape == (100, 3) and np.all(np.isfinite(sd))
    # empty dimensions, matrix input, weights keyword
    ds = Dataset(np.matrix([[1.0, 1.0, np.nan], [1.0, 1.0, np.nan]], dtype="float64"), weights=np.array([1.0, 2.0]))
    assert ds.empty_dimensions() == [2]
    # pickling a model built from a transposed matrix and a row mean
    m2 = PPCAModel(transform=np.matrix([[1, 1, 0], [1, 0, 1]], dtype="float64").T, isotropic_noise=0.1,
                   mean=np.array([[0, 1, 0]], dtype="float64"))
    de = pickle.loads(pickle.dumps(m2))
    assert repr(de) == repr(m2) and np.array_equal(de.transform, m2.transform)
    # priors
    prior = (Prior().with_isotropic_noise_prior(100.0, 100.0)
             .with_mean_prior(np.array([1.0, 0.0, 1.0], dtype="float64"), 0.0001 * np.eye(3, dtype="float64").T))
    s3 = m2.sample(100, mask_prob=0.2)
    pm = PPCAModel.init(2, s3)
    for _ in range(20):
        pm = pm.iterate_with_prior(s3, prior)
    assert np.abs(pm.to_canonical().mean - np.array([1.0, 0.0, 1.0])).max() < 0.05  # the tight mean prior wins
    # mixture: sample, train for several model counts, inference outputs
    mix = PPCAMix([PPCAModel(transform=np.matrix([[1, 0, 0], [0, 0, 1]], dtype="float64").T, isotropic_noise=0.1,
                             mean=np.array([[1, 1, 1]], dtype="float64").T),
                   PPCAModel(transform=np.matrix([[1, 1, 0], [1, 0, 1]], dtype="float64").T, isotropic_noise=0.1,
                             mean=np.array([[0, 1, 0]], dtype="float64").T)], log_weights=np.log([0.33333, 0.66667]))
    ms = mix.sample(100, 0.1)
    fitted = None
    for nm in (1, 2, 3):
        fitted = PPCAMixTrainer(ms).train(n_models=nm, state_size=2, n_iters=10, quiet=True)
    assert fitted.smooth(ms).numpy().shape == (100, 3) and fitted.extrapolate(ms).numpy().shape == (100, 3)
    assert fitted.infer(ms).posteriors().shape == (100, 3)
    # a 200 x 16 model (generic pipeline)
    big = PPCAModel(transform=np.matrix(np.random.default_rng(0).binomial(1.0, 0.1, size=(200, 16)), dtype="float64"),
                    isotropic_noise=0.1, mean=np.zeros((200, 1), dtype="float64"))
    bs = big.sample(20_000, 0.2)
    bm = PPCAModel.init(16, bs)
    vals = []
    for _ in range(4):
        vals.append(bm.llk(bs) / len(bs))
        bm = bm.iterate(bs)
    assert all(b >= a for a, b in zip(vals, vals[1:]))
    assert ppca_rs.__version__


def test_full_size_properties(P):
    """BASELINE config 2 at full size (N = 1M, d = 256, k = 10, 30 % masked): size-independent
    properties -- EM monotonicity, shard additivity of the statistics (the multi-GPU invariant),
    run-to-run bit reproducibility, extrapolate keeps observed entries."""
    from ppca_rs_amd import _lib

    n, d, k = 1_000_000, 256, 10
    rng = np.random.default_rng(12)
    truth = P.PPCAModel(0.1, rng.standard_normal((d, k)), rng.standard_normal(d))
    ds = truth.sample(n, 0.3, seed=1013)
    m = P.PPCAModel.init(k, ds, seed=2011)
    prev = -np.inf
    for _ in range(4):
        m, llk = m.iterate_with_llk(ds)
        assert llk >= prev
        prev = llk
    L = _lib.lib().ppca_stats_len(d, k)
    full = np.empty(L)
    again = np.empty(L)
    dev = m._device(ds._ctx)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, dev.h, _lib.ptr(full)))
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, dev.h, _lib.ptr(again)))
    np.testing.assert_array_equal(full, again)
    acc = np.zeros(L)
    for ch in ds.chunks(8):  # the 8-GPU sharding rule
        part = np.empty(L)
        _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ch._h, dev.h, _lib.ptr(part)))
        acc += part
    assert _rel(acc, full) < 1e-11
    sub = ds._slice(0, 4096)
    x = sub.numpy()
    ex = m.extrapolate(sub).numpy()
    ob = np.isfinite(x)
    np.testing.assert_array_equal(ex[ob], x[ob])
    assert np.isfinite(ex).all()


@pytest.mark.parametrize("n,d,k,mp,block", [(600, 300, 4, 0.3, False), (800, 200, 16, 0.2, False),
                                            (257, 1024, 64, 0.5, True), (90, 40, 12, 0.4, False),
                                            (300, 70, 20, 0.3, False), (200, 513, 10, 0.3, False),
                                            (301, 80, 40, 0.3, False), (203, 64, 32, 0.3, False), (7, 40, 17, 0.2, False),
                                            (130, 70, 48, 0.2, False), (66, 90, 49, 0.1, False),
                                            # both int8 products on the 256-row tile (>= 4096 samples, d >= 1024) with ODD numbers of
                                            # K-steps (17 over the dimensions, 67 over the samples): the wide request of the ring loop
                                            # walks K-steps in pairs
                                            (4230, 1080, 18, 0.3, False),
                                            # state sizes beyond 64 (round 5; the reference bounds k nowhere, ppca_model.rs:51-70): fp64
                                            # contractions + the workgroup-per-matrix solver of ppca_generic.hip, up to the 128 x 128
                                            # matrix that 160 KB of LDS hold
                                            (150, 90, 65, 0.2, False), (120, 140, 100, 0.3, False), (70, 150, 128, 0.2, False)])
def test_generic_pipeline_matches_oracle(P, oracle, n, d, k, mp, block):
    """Shapes outside the fused kernel (d > 256 or k > 10) run the split pipeline
    (ppca_generic.hip); BASELINE config 4 (d = 1024, k = 64, 50 % block-masked) at oracle-sized N."""
    from ppca_rs_amd import _lib

    assert _lib.lib().ppca_path_kind(d, k) == 0
    rng = np.random.default_rng(d + k)
    if block:  # one cyclic run of d/2 masked dims per sample (cfg 4)
        x, _, _ = oracle.synth(n, d, k, 0.0, 700 + d)
        for i in range(n):
            st = rng.integers(0, d)
            x[i, (st + np.arange(d // 2)) % d] = np.nan
    else:
        x, _, _ = oracle.synth(n, d, k, mp, 700 + d)
    x[1] = np.nan
    w = rng.uniform(0.5, 1.5, n)
    c, mu, s = 0.3 * rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), 0.9
    ds, m = P.Dataset(x, w), P.PPCAModel(s, c, mu)
    L = _lib.lib().ppca_stats_len(d, k)
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    want = oracle.stats(x, s, c, mu, w)
    kp = k * (k + 1) // 2
    bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
    for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
        assert _rel(got[a:b], want[a:b]) < 1e-8, name
    assert _rel(m.llks(ds), oracle.llks(x, s, c, mu)) < 1e-9
    assert abs(m.llk(ds) - oracle.llk(x, s, c, mu, w)) < 1e-9 * abs(oracle.llk(x, s, c, mu, w))
    inf = m.infer(ds)
    st, cv = oracle.infer(x, s, c, mu)
    assert _rel(inf.states(), st) < 1e-7 and _rel(np.array(inf.covariances()), cv) < 1e-7
    assert _rel(m.smooth(ds).numpy(), oracle.reconstruct(x, s, c, mu, "smooth")) < 1e-8
    ex = m.extrapolate(ds).numpy()
    assert _rel(ex, oracle.reconstruct(x, s, c, mu, "extrapolate")) < 1e-8
    np.testing.assert_array_equal(ex[np.isfinite(x)], x[np.isfinite(x)])
    for _ in range(2):
        want_llk = oracle.llk(x, s, c, mu, w)
        s, c, mu = oracle.iterate(x, s, c, mu, w)
        m, llk = m.iterate_with_llk(ds)
        assert abs(llk - want_llk) < RTOL * 1e-3 * abs(want_llk)
        assert abs(m.isotropic_noise - s) < RTOL * s
        assert _rel(m.transform, c) < RTOL and _rel(m.mean, mu) < RTOL
    diag = m.infer(ds).smoothed_covariances_diagonal(m).numpy()
    _lib_h = C.c_void_p()
    _lib.check(_lib.lib().ppca_covariance_diagonal(ds._ctx.handle, ds._h, m._device(ds._ctx).h, 0, C.byref(_lib_h)))
    dev_diag = P.Dataset._wrap(_lib_h, ds._ctx).numpy()
    assert _rel(dev_diag, diag) < 1e-7


def test_last_guard_after_a_larger_model(P, oracle):
    """ppca_em_last_guard reads the first four Gram-guard flags whatever the state size of the last pass: after a pass with more
    packed-column tiles (k = 10: four; the two-kernel pass and the split pipeline share the buffer) a k = 1 pass must not report
    the other model's flags (found by running the split-pipeline tests right before the steady-state tests of k = 1, 4, 7)."""
    from ppca_rs_amd import _lib

    ctx = _lib.default_context()
    rng = np.random.default_rng(77)
    for k_big, d_big in ((10, 256), (16, 200), (20, 70)):
        xb, _, _ = oracle.synth(300, d_big, k_big, 0.2, 60 + k_big)
        cb = rng.standard_normal((d_big, k_big))
        cb[0] *= 1e8  # a model whose Gram guard trips (rows of C spanning 1e8): its flags stay in the buffer
        P.PPCAModel(1.0, cb, np.zeros(d_big)).iterate(P.Dataset(xb))
        for k, d in ((1, 64), (4, 200), (7, 255)):
            x, _, _ = oracle.synth(500, d, k, 0.3, 80 + k)
            m = P.PPCAModel(0.7, 0.6 * rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d))
            m.iterate(P.Dataset(x))
            assert ctx.last_guard() == (0, 0), (k_big, k)


def test_two_ranks_on_one_gpu_match_single_rank(P, tmp_path):
    """The N > 1 path of bench.py (row shards, per-rank pass, all-reduce of the statistics buffer on the
    launch stream, replicated finalisation) with two ranks sharing this GPU over gloo: the model after
    a few EM steps equals the single-rank one.  RCCL itself is only exercised by the driver's 8-GPU run."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--samples", "200000", "--no-cpu"]
    one = str(tmp_path / "one.npz")
    two = str(tmp_path / "two.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dump-model", one] + common,
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--dump-model", two] + common,
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert abs(a["sigma"] - b["sigma"]) < 1e-10 * a["sigma"]
    assert _rel(b["transform"], a["transform"]) < 1e-9 and _rel(b["mean"], a["mean"]) < 1e-9
    assert abs(a["llk"] - b["llk"]) < 1e-10 * abs(a["llk"])


@pytest.mark.parametrize("shape", ["small", "cfg5"])
def test_sharded_mixture_two_ranks_on_one_gpu(shape):
    """BASELINE config 5's multi-GPU path (ShardedMixEM: all-reduce MAX of the component maxima, one all-reduce SUM
    of K statistic buffers + weight sums + llk) with two gloo ranks sharing this GPU, against the single-process
    PPCAMix.iterate on the whole dataset -- at a small shape and at config 5's own (8 components, d = 256, k = 10,
    weighted; there also against the committed oracle fixture)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541" if shape == "small" else "29543",
                        os.path.join(root, "tools", "mix_sharded_check.py")] + ([] if shape == "small" else ["cfg5"]),
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "sharded mixture OK" in r.stdout, (r.stdout[-1500:], r.stderr[-2500:])


# --------------------------------------------------------------------------- round 2
def _cfg5_fixture():
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from inputs import cfg5_inputs, digest

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg5_d256_k10_m8.npz"))
    inp = cfg5_inputs()
    assert digest(*inp) == str(g["digest"])
    return g, inp


def test_config5_shape_mixture(P):
    """BASELINE config 5 at ITS shape -- 8 components, d = 256, state_size = 10, 30 % masked, non-uniform sample
    weights (the weighted branch of pass_kernel<10, ...> with the per-sample logarithm, and the no_llk component
    steps) -- against the committed oracle fixture: mixture llks, log posteriors, two weighted EM iterations."""
    g, (x, w, sig, cs, ms, lw) = _cfg5_fixture()
    ds = P.Dataset(x, w)
    mix = P.PPCAMix([P.PPCAModel(sig[c], cs[c], ms[c]) for c in range(len(sig))], lw)
    assert _rel(mix.llks(ds), g["llks"]) < 1e-10
    lp = mix.infer_cluster(ds)
    assert np.abs(np.exp(lp) - np.exp(g["log_posterior"])).max() < 1e-9
    soft = np.exp(g["log_posterior"]).max(axis=1) < 0.99
    assert soft.mean() > 0.05 and np.abs(lp[soft] - g["log_posterior"][soft]).max() < 1e-7
    for it in range(len(g["it_llk"])):
        mix, llk = mix.iterate_with_llk(ds)
        assert abs(llk - g["it_llk"][it]) < 1e-9 * abs(g["it_llk"][it]), it
        for c, mdl in enumerate(mix.models):
            assert abs(mdl.isotropic_noise - g["it_sigma"][it][c]) < RTOL * 1e-2 * g["it_sigma"][it][c], (it, c)
            assert _rel(mdl.transform, g["it_c"][it][c]) < RTOL * 1e-1, (it, c)
            assert _rel(mdl.mean, g["it_mean"][it][c]) < RTOL * 1e-1, (it, c)
        assert np.abs(np.exp(mix.log_weights) - np.exp(g["it_lw"][it])).max() < 1e-9, it


def _engine(P, ds, m):
    from ppca_rs_amd import _lib

    e = C.c_int32(-1)
    _lib.check(_lib.lib().ppca_gram_engine(ds._ctx.handle, m._device(ds._ctx).h, C.byref(e)))
    return e.value


def test_int8_gram_dynamic_range_guard(P, oracle):
    """The int8-sliced Gram keeps 62 bits below each column maximum of vech(c c^T); when a sample masks the rows that
    set the maximum, its Gram would be summed from truncated entries.  qprep's guard must send such models to the
    fp64-MFMA Gram on the device (output_covariance.rs:57-70 computes C_o^T C_o in f64), and keep the benign ones on
    the int8 engine."""
    rng = np.random.default_rng(77)
    d, k, n = 256, 10, 640
    big = np.sort(rng.choice(d, 200, replace=False))
    small = np.setdiff1d(np.arange(d), big)
    # benign model of the headline shape: int8
    c0, mu = rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d)
    x0, _, _ = oracle.synth(n, d, k, 0.3, 4242)
    assert _engine(P, P.Dataset(x0), P.PPCAModel(0.1, c0, mu)) == 0
    # (a) rows of C spanning 1e8, the 200 large ones masked in EVERY sample (a model applied to data without those
    # dims): every sample's Gram is O(1) under a column maximum of O(1e17) -- all digits of the int8 form fall off.
    # The oracle is meaningful here (|G| / sigma^2 is moderate): statistics, llks, posteriors, one EM step.
    scale = np.ones(d)
    scale[big] = 1.0e8
    c = rng.standard_normal((d, k)) * scale[:, None]
    s = 0.05
    z = rng.standard_normal((n, k))
    x = z @ c.T + mu + s * rng.standard_normal((n, d))
    x[rng.random((n, d)) < 0.3] = np.nan
    xa = x.copy()
    xa[:, big] = np.nan
    w = rng.uniform(0.5, 1.5, n)
    ds, m = P.Dataset(xa, w), P.PPCAModel(s, c, mu)
    assert _engine(P, ds, m) == 1
    from ppca_rs_amd import _lib

    L = _lib.lib().ppca_stats_len(d, k)
    got = np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, m._device(ds._ctx).h, _lib.ptr(got)))
    want = oracle.stats(xa, s, c, mu, w)
    kp = k * (k + 1) // 2
    bounds = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d, L]
    for name, a, b in zip(["cross", "S", "U", "sumx", "totals", "scalars"], bounds[:-1], bounds[1:]):
        assert _rel(got[a:b], want[a:b]) < 1e-9, name
    assert _rel(m.llks(ds), oracle.llks(xa, s, c, mu)) < 1e-9
    st, cv = oracle.infer(xa, s, c, mu)
    inf = m.infer(ds)
    # (the oracle forms Sigma subtractively, I - C_o^T(...)C_o-style: entries of 1e-4 out of 1 - (1 - 1e-4))
    assert _rel(inf.states(), st) < 1e-9 and _rel(np.array(inf.covariances()), cv) < 1e-6
    new, llk = m.iterate_with_llk(ds)
    s1, c1, m1 = oracle.iterate(xa, s, c, mu, w)
    assert abs(llk - oracle.llk(xa, s, c, mu, w)) < 1e-9 * abs(llk)
    assert abs(new.isotropic_noise - s1) < 1e-8 * s1 and _rel(new.transform[small], c1[small]) < 1e-8
    np.testing.assert_array_equal(new.transform[big], c[big])  # empty dimensions keep their rows
    # (b) the large rows masked in HALF the samples: posterior means against a long-double dense evaluation (the
    # oracle's subtractive Woodbury form is void at |G| / sigma^2 ~ 1e21)
    xb = x.copy()
    half = np.arange(n) % 2 == 0
    xb[np.ix_(half, big)] = np.nan
    dsb = P.Dataset(xb)
    assert _engine(P, dsb, m) == 1
    zs = m.infer(dsb).states()
    ld = np.longdouble
    worst = 0.0
    for i in list(range(0, 40)) + list(range(n - 40, n)):
        o = np.isfinite(xb[i])
        co = c[o].astype(ld)
        M = co.T @ co + ld(s) ** 2 * np.eye(k, dtype=ld)
        b = co.T @ (xb[i, o] - mu[o]).astype(ld)
        Lc = np.zeros((k, k), dtype=ld)  # long-double Cholesky (numpy.linalg has no extended precision)
        for a in range(k):
            for bb in range(a + 1):
                v = M[a, bb] - (Lc[a, :bb] * Lc[bb, :bb]).sum()
                Lc[a, bb] = np.sqrt(v) if a == bb else v / Lc[bb, bb]
        y = np.zeros(k, dtype=ld)
        for a in range(k):
            y[a] = (b[a] - (Lc[a, :a] * y[:a]).sum()) / Lc[a, a]
        zz = np.zeros(k, dtype=ld)
        for a in reversed(range(k)):
            zz[a] = (y[a] - (Lc[a + 1:, a] * zz[a + 1:]).sum()) / Lc[a, a]
        worst = max(worst, float(np.abs(zs[i] - zz.astype(np.float64)).max() / max(np.abs(zz).max(), 1e-300)))
    assert worst < 1e-9, worst
    # (c) a non-finite product cannot be carried by the fixed-point form: fp64 engine, and the NaN propagates
    cn = c0.copy()
    cn[3, 2] = np.inf
    mn = P.PPCAModel(0.5, cn, mu)
    assert _engine(P, P.Dataset(x0), mn) == 1
    assert not np.isfinite(mn.llk(P.Dataset(x0)))
    # (d) what a trained model looks like -- sigma = 1e-3 of the scale and one weakly loaded dimension (row norm 1e-2 of
    # the others) -- stays on the int8 engine (8-bit digits: 62 bits under each column maximum; the 7-bit digits of rounds
    # 1-2 sent it to the fp64 engine) and agrees with a long-double dense evaluation of the posterior mean and the
    # log-likelihood (the oracle's subtractive Woodbury form has ~7 digits left at |G| / sigma^2 ~ 2e9)
    cd = rng.standard_normal((d, k))
    cd[5] *= 1.0e-2
    sd = 1.0e-3
    xd = rng.standard_normal((n, k)) @ cd.T + mu + sd * rng.standard_normal((n, d))
    xd[rng.random((n, d)) < 0.3] = np.nan
    md, dsd = P.PPCAModel(sd, cd, mu), P.Dataset(xd)
    assert _engine(P, dsd, md) == 0
    zs, lls = md.infer(dsd).states(), md.llks(dsd)
    ld = np.longdouble
    wz = wl = 0.0
    for i in list(range(0, 30)) + list(range(n - 30, n)):
        o = np.isfinite(xd[i])
        co = cd[o].astype(ld)
        M = co.T @ co + ld(sd) ** 2 * np.eye(k, dtype=ld)
        xt = (xd[i, o] - mu[o]).astype(ld)
        b = co.T @ xt
        Lc = np.zeros((k, k), dtype=ld)
        for a in range(k):
            for bb in range(a + 1):
                v = M[a, bb] - (Lc[a, :bb] * Lc[bb, :bb]).sum()
                Lc[a, bb] = np.sqrt(v) if a == bb else v / Lc[bb, bb]
        y = np.zeros(k, dtype=ld)
        for a in range(k):
            y[a] = (b[a] - (Lc[a, :a] * y[:a]).sum()) / Lc[a, a]
        zz = np.zeros(k, dtype=ld)
        for a in reversed(range(k)):
            zz[a] = (y[a] - (Lc[a + 1:, a] * zz[a + 1:]).sum()) / Lc[a, a]
        mo = int(o.sum())
        # (|x~|^2 - b^T M^-1 b) / s^2 = |x~ - C_o z|^2 / s^2 + |z|^2 without the cancellation (ppca_model.rs:124-139)
        r = xt - co @ zz
        llk_i = -ld(0.5) * ((r * r).sum() / ld(sd) ** 2 + (zz * zz).sum() + 2 * np.log(np.diag(Lc)).sum()
                            + 2 * np.log(ld(sd)) * (mo - k) + np.log(2 * ld(np.pi)) * mo)
        wz = max(wz, float(np.abs(zs[i] - zz.astype(np.float64)).max() / max(np.abs(zz).max(), 1e-300)))
        wl = max(wl, abs(float(lls[i]) - float(llk_i)) / abs(float(llk_i)))
    # (the llk's (|x~|^2 - b^T M^-1 b) / sigma^2 is a difference of two numbers ~1e7 times its size at this sigma, in the
    #  reference's quadratic_form as here: 1e-7 is what fp64 leaves of it)
    assert wz < 1e-9 and wl < 1e-7, (wz, wl)


def test_library_rccl_communicator_single_rank(P, oracle):
    """The collective behind the C-ABI (ppca_comm_*, ppca_em_step_sharded, ppca_em_step_group) through RCCL with a
    clique of ONE rank -- the only size a 1-GPU box admits (RCCL refuses two ranks on one device): the sharded step
    must equal the plain one bit for bit, the sum / max all-reduces must be identities."""
    import torch

    from ppca_rs_amd import _lib
    from ppca_rs_amd.distributed import Communicator, ShardedEM

    assert Communicator.backend().startswith("rccl"), Communicator.backend()
    ctx = _lib.default_context()
    comm = Communicator(ctx, 1, 0, Communicator.unique_id())
    x, _, _ = oracle.synth(3000, 256, 10, 0.3, 91)
    rng = np.random.default_rng(2)
    start = P.PPCAModel(1.0, rng.standard_normal((256, 10)), np.zeros(256))
    ds = P.Dataset(x)
    want, want_llk = start.iterate_with_llk(ds)
    em = ShardedEM(ds, start, comm=comm)
    em.step()
    got = em.model()
    assert em.llk_of_previous() == want_llk
    np.testing.assert_array_equal(got.transform, want.transform)
    np.testing.assert_array_equal(got.mean, want.mean)
    assert got.isotropic_noise == want.isotropic_noise
    em.step()
    want2 = want.iterate(ds)
    np.testing.assert_array_equal(em.model().transform, want2.transform)
    em.close()
    t = torch.arange(1000, dtype=torch.float64, device="cuda") - 3.5
    for op in ("sum", "max"):
        u = t.clone()
        comm.allreduce(u.data_ptr(), u.numel(), op)
        ctx.synchronize()
        assert torch.equal(u, t)
    # the sharded mixture step with both of its all-reduces (MAX of the component maxima, SUM of the statistics) inside
    # the library: a clique of one must reproduce PPCAMix.iterate
    from ppca_rs_amd.distributed import ShardedMixEM

    xm = np.concatenate([oracle.synth(400, 24, 3, 0.25, 300 + c_, mean_scale=3.0)[0] for c_ in range(3)])
    mrng = np.random.default_rng(9)
    mix0 = P.PPCAMix([P.PPCAModel(1.0, mrng.standard_normal((24, 3)), mrng.standard_normal(24)) for _ in range(3)],
                     np.log([0.3, 0.3, 0.4]))
    dsm = P.Dataset(xm, mrng.uniform(0.5, 1.5, xm.shape[0]))
    want_mix, want_mllk = mix0.iterate_with_llk(dsm)
    sm = ShardedMixEM(dsm, mix0, comm=comm)
    got_mllk = sm.step()
    got_mix = sm.mixture()
    assert abs(got_mllk - want_mllk) < 1e-12 * abs(want_mllk)
    assert _rel(got_mix.log_weights, want_mix.log_weights) < 1e-12
    for ga, wa in zip(got_mix.models, want_mix.models):
        assert _rel(ga.transform, wa.transform) < 1e-12 and abs(ga.isotropic_noise - wa.isotropic_noise) < 1e-13
    # one thread, a group of one device
    lib = _lib.lib()
    comms = (C.c_void_p * 1)()
    ctxs = (C.c_void_p * 1)(ctx.handle)
    _lib.check(lib.ppca_comm_create_all(ctxs, 1, comms))
    out = C.c_void_p()
    _lib.check(lib.ppca_model_alloc(ctx.handle, 256, 10, C.byref(out)))
    llk = C.c_double(0.0)
    shards, mins, mouts = (C.c_void_p * 1)(ds._h), (C.c_void_p * 1)(start._device(ctx).h), (C.c_void_p * 1)(out)
    _lib.check(lib.ppca_em_step_group(comms, 1, shards, mins, None, mouts, C.byref(llk)))
    sig = C.c_double(0.0)
    cc, mm = np.empty((256, 10)), np.empty(256)
    _lib.check(lib.ppca_model_download(out, C.byref(sig), _lib.ptr(cc), _lib.ptr(mm)))
    assert llk.value == want_llk and sig.value == want.isotropic_noise
    np.testing.assert_array_equal(cc, want.transform)
    lib.ppca_model_free(out)
    lib.ppca_comm_destroy(comms[0])
    comm.close()


def test_rccl_two_ranks_when_two_gpus(tmp_path):
    """bench.py --gpus 2 started as a plain process: it must spawn its own ranks, run the all-reduce over RCCL
    (the library's communicator), and reach the single-rank model.  Skipped on a 1-GPU box."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--samples", "200000", "--no-cpu"]
    one, two, three = (str(tmp_path / f) for f in ("one.npz", "two.npz", "three.npz"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for gpus, dump, extra in ((1, one, []), (2, two, ["--collective", "capi"]), (2, three, ["--collective", "torch"])):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--dump-model", dump] + common + extra,
                           capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        import json

        assert json.loads(line)["n_gpus"] == gpus
    a = np.load(one)
    for other in (two, three):
        b = np.load(other)
        assert abs(a["sigma"] - b["sigma"]) < 1e-10 * a["sigma"] and _rel(b["transform"], a["transform"]) < 1e-9


def test_group_step_two_devices_when_two_gpus(P, oracle):
    """ppca_comm_create_all + ppca_em_step_group with n = 2: ONE host thread drives two GPUs, the two all-reduces issued as one
    RCCL group; both devices must end on the model a single device reaches on the whole dataset.  Skipped on a 1-GPU box."""
    import torch

    from ppca_rs_amd import _lib

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    lib = _lib.lib()
    x, _, _ = oracle.synth(6001, 256, 10, 0.3, 17)
    rng = np.random.default_rng(4)
    start = P.PPCAModel(1.0, rng.standard_normal((256, 10)), np.zeros(256))
    ctx0 = _lib.default_context()
    want, want_llk = start.iterate_with_llk(P.Dataset(x))
    ctxs = [_lib.Context(0), _lib.Context(1)]
    half = (len(x) + 1) // 2  # the rule of Dataset.chunks (src/python_bindings.rs:110-118)
    shards = [P.Dataset(x[:half], ctx=ctxs[0]), P.Dataset(x[half:], ctx=ctxs[1])]
    comms = (C.c_void_p * 2)()
    _lib.check(lib.ppca_comm_create_all((C.c_void_p * 2)(ctxs[0].handle, ctxs[1].handle), 2, comms))
    assert lib.ppca_comm_n_ranks(comms[0]) == 2 and lib.ppca_comm_rank(comms[1]) == 1
    outs = []
    for c in ctxs:
        h = C.c_void_p()
        _lib.check(lib.ppca_model_alloc(c.handle, 256, 10, C.byref(h)))
        outs.append(h)
    llk = C.c_double(0.0)
    start1 = P.PPCAModel(1.0, start.transform, start.mean)  # (a model object caches ONE device copy)
    ins = [start._device(ctxs[0]), start1._device(ctxs[1])]
    _lib.check(lib.ppca_em_step_group(comms, 2, (C.c_void_p * 2)(shards[0]._h, shards[1]._h), (C.c_void_p * 2)(ins[0].h, ins[1].h),
                                      None, (C.c_void_p * 2)(*outs), C.byref(llk)))
    assert abs(llk.value - want_llk) < 1e-11 * abs(want_llk)
    got = []
    for h in outs:
        sig, cc, mm = C.c_double(0.0), np.empty((256, 10)), np.empty(256)
        _lib.check(lib.ppca_model_download(h, C.byref(sig), _lib.ptr(cc), _lib.ptr(mm)))
        got.append((sig.value, cc, mm))
        lib.ppca_model_free(h)
    assert got[0][0] == got[1][0]
    np.testing.assert_array_equal(got[0][1], got[1][1])  # identical finalisation on both devices, no broadcast
    np.testing.assert_array_equal(got[0][2], got[1][2])
    assert abs(got[0][0] - want.isotropic_noise) < 1e-10 * want.isotropic_noise and _rel(got[0][1], want.transform) < 1e-9
    for i in range(2):
        lib.ppca_comm_destroy(comms[i])
    del ctx0


def test_sharded_mixture_over_rccl_when_two_gpus():
    """BASELINE config 5's step over two row shards with BOTH of its collectives inside the library
    (ppca_mix_em_step_sharded: all-reduce(MAX) of the component maxima, all-reduce(SUM) of [K statistics | K sums | llk]):
    bench.py --config 5 --gpus 2 must trace the log-likelihoods of the one-GPU run.  Skipped on a 1-GPU box."""
    import json
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    traces = []
    for gpus in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "5", "--gpus", str(gpus), "--samples", "400000",
                            "--components", "4", "--steps", "3", "--warmup", "3", "--no-cpu"] + (["--collective", "capi"] if gpus > 1 else []),
                           capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["n_gpus"] == gpus
        if gpus > 1:
            assert "ppca_mix_em_step_sharded" in j["config"]["collective"]
        traces.append(np.array(j["llk_per_sample_trace"]))
    assert _rel(traces[1], traces[0]) < 1e-9


def test_concurrent_calls_on_one_context(P, oracle):
    """The reference's methods release the GIL (src/python_bindings.rs:466-511) and may be called from several
    threads; ctypes drops the GIL too, so four threads hammer ONE context with different models (llk, iterate,
    infer, extrapolate) and must get exactly the serial results."""
    import threading

    x, _, _ = oracle.synth(20000, 256, 10, 0.3, 17)
    ds = P.Dataset(x)
    rng = np.random.default_rng(3)
    models = [P.PPCAModel(0.5 + 0.2 * i, rng.standard_normal((256, 10)), 0.1 * rng.standard_normal(256)) for i in range(4)]

    def work(m):
        new, llk = m.iterate_with_llk(ds)
        return (m.llk(ds), llk, new.transform.copy(), new.isotropic_noise, m.infer(ds).states()[:50].copy(),
                m.extrapolate(ds._slice(0, 64)).numpy())

    serial = [work(m) for m in models]
    results, errors = [None] * 4, []

    def run(i):
        try:
            for _ in range(6):
                results[i] = work(models[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for got, want in zip(results, serial):
        assert got[0] == want[0] and got[1] == want[1] and got[3] == want[3]
        for a, b in zip((got[2], got[4], got[5]), (want[2], want[4], want[5])):
            np.testing.assert_array_equal(a, b)


def test_mixture_with_different_state_sizes(P):
    """mix.rs:50-71 admits components of different state sizes (state_sizes, :91): k = (2, 4, 12) here -- two fused
    instantiations and the generic pipeline in one mixture -- against the second restatement
    (oracle/restate_numpy.py, mix.rs:281-337), weighted."""
    from oracle import restate_numpy as R

    rng = np.random.default_rng(23)
    n, d, ks = 160, 14, (2, 4, 12)
    x = rng.standard_normal((n, 3)) @ rng.standard_normal((3, d)) + 0.4 * rng.standard_normal((n, d))
    x[rng.random((n, d)) < 0.25] = np.nan
    w = rng.uniform(0.5, 1.5, n)
    sig = [0.9, 1.1, 0.7]
    cs = [0.6 * rng.standard_normal((d, k)) for k in ks]
    ms = [0.3 * rng.standard_normal(d) for _ in ks]
    lw = np.log(np.array([0.2, 0.5, 0.3]))
    ds = P.Dataset(x, w)
    mix = P.PPCAMix([P.PPCAModel(s, c, m) for s, c, m in zip(sig, cs, ms)], lw)
    assert mix.state_sizes == list(ks)
    comp = np.stack([R.llks(x, s, c, m) for s, c, m in zip(sig, cs, ms)], axis=1) + lw
    want_llks = np.log(np.exp(comp - comp.max(1, keepdims=True)).sum(1)) + comp.max(1)
    assert _rel(mix.llks(ds), want_llks) < 1e-9
    for _ in range(2):
        s1, c1, m1, lw1 = R.mix_iterate(x, np.array(sig), cs, ms, lw, w)
        mix = mix.iterate(ds)
        for c_, mdl in enumerate(mix.models):
            assert abs(mdl.isotropic_noise - s1[c_]) < 1e-7 * s1[c_]
            assert _rel(mdl.transform, c1[c_]) < 1e-6 and _rel(mdl.mean, m1[c_]) < 1e-6
        assert _rel(mix.log_weights, lw1) < 1e-8
        sig, cs, ms, lw = list(s1), c1, m1, lw1
    assert mix.smooth(ds).numpy().shape == x.shape and mix.infer(ds).posteriors().shape == (n, 3)


def test_full_size_properties_config4(P):
    """BASELINE config 4 at full size (N = 2M, d = 1024, k = 64, one cyclic run of d/2 masked dims per sample; the
    generic pipeline with both int8-sliced contractions and the MFMA-blocked inversion): size-independent properties
    -- EM monotonicity, shard additivity of the statistics, run-to-run bit reproducibility, extrapolate keeps observed
    entries -- and the int8 engine against the fp64 GEMMs on a slice (PPCA_GENERIC_FP64 is read once per process, so
    the comparison is between the shard sums and a finer sharding, which take different chunkings)."""
    from ppca_rs_amd import _lib

    n, d, k = 2_000_000, 1024, 64
    ctx = _lib.default_context()
    truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
    spec = _lib.SynthSpec(0, n, d, k, 0.1, 0.0, 1, d // 2, 1033, truth._c.ctypes.data_as(_lib.c_double_p),
                          truth._mean.ctypes.data_as(_lib.c_double_p))
    h = C.c_void_p()
    _lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
    ds = P.Dataset._wrap(h, ctx)
    m = P.PPCAModel.init(k, ds, seed=3)
    prev = -np.inf
    for _ in range(3):
        m, llk = m.iterate_with_llk(ds)
        assert np.isfinite(llk) and llk >= prev
        prev = llk
    sub = ds._slice(0, 300_000)
    L = _lib.lib().ppca_stats_len(d, k)
    dev = m._device(ctx)
    full, again, acc = np.empty(L), np.empty(L), np.zeros(L)
    _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, sub._h, dev.h, _lib.ptr(full)))
    _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, sub._h, dev.h, _lib.ptr(again)))
    np.testing.assert_array_equal(full, again)
    for ch in sub.chunks(8):
        part = np.empty(L)
        _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ch._h, dev.h, _lib.ptr(part)))
        acc += part
    assert _rel(acc, full) < 1e-10
    x = ds._slice(0, 512).numpy()
    assert abs(np.isnan(x).mean() - 0.5) < 0.01  # half of every row masked
    ex = m.extrapolate(ds._slice(0, 512)).numpy()
    ob = np.isfinite(x)
    np.testing.assert_array_equal(ex[ob], x[ob])
    assert np.isfinite(ex).all()


def test_full_size_properties_reference_largest_workload(P):
    """The reference's own largest workload shape (lib.rs:82-99: d = 200, k = 16) at N = 2.5 M (three chunks of the
    two-kernel pass, ppca_em16.hip): size-independent properties -- EM monotonicity; the integer path of the
    statistics kernel BIT-EXACT (totals = per-dimension observed counts, which a column scan of the data gives
    independently); sumx against a float64 column sum; shard additivity; run-to-run bit reproducibility."""
    from ppca_rs_amd import _lib

    n, d, k = 2_500_000, 200, 16
    ctx = _lib.default_context()
    truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
    spec = _lib.SynthSpec(0, n, d, k, 0.1, 0.3, 0, 0, 1033, truth._c.ctypes.data_as(_lib.c_double_p),
                          truth._mean.ctypes.data_as(_lib.c_double_p))
    h = C.c_void_p()
    _lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
    ds = P.Dataset._wrap(h, ctx)
    m = P.PPCAModel.init(k, ds, seed=3)
    prev = -np.inf
    for _ in range(3):
        m, llk = m.iterate_with_llk(ds)
        assert np.isfinite(llk) and llk >= prev
        prev = llk
    L = _lib.lib().ppca_stats_len(d, k)
    kp = k * (k + 1) // 2
    o_sumx, o_tot = 2 * d * k + d * kp, 2 * d * k + d * kp + d
    dev = m._device(ctx)
    full, again = np.empty(L), np.empty(L)
    _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ds._h, dev.h, _lib.ptr(full)))
    _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ds._h, dev.h, _lib.ptr(again)))
    np.testing.assert_array_equal(full, again)
    counts, sums = np.zeros(d), np.zeros(d)
    for ch in ds.chunks(10):  # (250 000 x 200 doubles per download)
        x = ch.numpy()
        ob = np.isfinite(x)
        counts += ob.sum(axis=0)
        sums += np.where(ob, x - m.mean, 0.0).sum(axis=0)
    np.testing.assert_array_equal(full[o_tot:o_tot + d], counts)  # exact: sums of 0/1 x weight 1 through int64
    assert _rel(full[o_sumx:o_sumx + d], sums) < 1e-10
    acc = np.zeros(L)
    for ch in ds.chunks(3):
        part = np.empty(L)
        _lib.check(_lib.lib().ppca_stats_raw(ctx.handle, ch._h, dev.h, _lib.ptr(part)))
        acc += part
    assert _rel(acc, full) < 1e-10


def test_full_size_properties_config5(P):
    """BASELINE config 5 at full size on one GPU (8 components, N = 5M, d = 256, k = 10, 30 % masked): the mixture
    log-likelihood never decreases over EM iterations (mix.rs:281-337 is an EM step), the weights stay normalised,
    and the step is reproducible bit for bit (deterministic row selection and reductions)."""
    n, d, k, nm = 5_000_000, 256, 10, 8
    rng = np.random.default_rng(1)
    parts = [P.PPCAModel(0.1, rng.standard_normal((d, k)), 3.0 * rng.standard_normal(d)).sample(n // nm, 0.3, seed=100 + c)
             for c in range(nm)]
    ds = P.Dataset.concat(parts)
    del parts
    mix = P.PPCAMix.init(nm, k, ds, seed=7)
    prev = -np.inf
    for it in range(6):
        new, llk = mix.iterate_with_llk(ds)
        assert np.isfinite(llk) and llk >= prev - 1e-9 * abs(llk), it
        prev = llk
        if it == 4:
            again, llk2 = mix.iterate_with_llk(ds)
            assert llk2 == llk
            for a, b in zip(new.models, again.models):
                np.testing.assert_array_equal(a.transform, b.transform)
            np.testing.assert_array_equal(new.log_weights, again.log_weights)
        mix = new
        assert abs(np.exp(mix.log_weights).sum() - 1.0) < 1e-12


@pytest.mark.gpu
def test_block_cache_reuses_and_trims(P):
    """Output datasets come from the context's block cache (ppca_ctx_trim, include/ppca_hip.h): a released block is
    handed to the next allocation of its size, results do not depend on what the block held before, and trim gives
    the memory back.  A private context so that other tests' blocks do not enter the count."""
    from ppca_rs_amd import _lib
    ctx = _lib.Context(0)
    rng = np.random.default_rng(5)
    n, d, k = 40_000, 64, 5
    truth = P.PPCAModel(0.1, rng.standard_normal((d, k)), rng.standard_normal(d))
    ds = P.Dataset(truth.sample(n, 0.3, seed=3).numpy(), ctx=ctx)
    model = P.PPCAModel.init(k, ds, seed=1).iterate(ds)
    assert ctx.trim() >= 0
    first = model.smooth(ds)
    want = first.numpy()
    del first                                   # the N x d block goes back to the cache ...
    junk = model.extrapolate(ds)                # ... is drawn and dirtied by another pass ...
    del junk
    again = model.smooth(ds)                    # ... and drawn again
    np.testing.assert_array_equal(again.numpy(), want)
    del again
    released = ctx.trim()
    assert released >= n * d * 8, released
    assert ctx.trim() == 0
    np.testing.assert_array_equal(model.smooth(ds).numpy(), want)


@pytest.mark.gpu
def test_state_size_zero_is_the_isotropic_model(P):
    """state_size = 0 (ppca_model.rs:51-70 with an empty transform, to_canonical :399-402): an isotropic Gaussian around
    the mean.  Checked against the closed forms: llk_i = -1/2 [|x~_i|^2 / s^2 + 2 m_i ln s + m_i ln 2 pi]; smooth = mean;
    extrapolate keeps the observed values; iterate moves only the mean and sigma:
    mean_j += sum_i w_i m_ij x~_ij / sum_i w_i m_ij,  sigma^2 = sum_i w_i |x~_i|^2 / sum_ij w_i m_ij  (:328-377 with no
    transform: the trace term and C z vanish; the noise sums use the OLD mean)."""
    rng = np.random.default_rng(11)
    n, d = 700, 40
    x = 0.7 * rng.standard_normal((n, d)) + rng.standard_normal(d)
    x[rng.random((n, d)) < 0.3] = np.nan
    x[5] = np.nan  # an all-masked sample
    w = rng.uniform(0.5, 2.0, n)
    mu, s = rng.standard_normal(d), 0.8
    m = P.PPCAModel(s, np.zeros((d, 0)), mu)
    assert m.state_size == 0 and m.n_parameters == 1 + d and m.to_canonical() is m
    ds = P.Dataset(x, w)
    ob = np.isfinite(x)
    xt = np.where(ob, x - mu, 0.0)
    mi = ob.sum(axis=1)
    llks = -0.5 * ((xt ** 2).sum(axis=1) / s ** 2 + 2 * mi * np.log(s) + mi * np.log(2 * np.pi))
    llks[mi == 0] = 0.0
    assert _rel(m.llks(ds), llks) < 1e-12
    assert abs(m.llk(ds) - (w * llks).sum()) < 1e-10 * abs((w * llks).sum())
    np.testing.assert_allclose(m.smooth(ds).numpy(), np.tile(mu, (n, 1)), rtol=0, atol=1e-15)
    ex = m.extrapolate(ds).numpy()
    np.testing.assert_array_equal(ex[ob], x[ob])
    np.testing.assert_allclose(ex[~ob], np.tile(mu, (n, 1))[~ob], rtol=0, atol=1e-15)
    inf = m.infer(ds)
    assert inf.states().shape == (n, 0) and len(inf.covariances()) == n
    new, llk = m.iterate_with_llk(ds)
    assert new.state_size == 0 and new.transform.shape == (d, 0)
    tot = (w[:, None] * ob).sum(axis=0)
    mean_want = mu + (w[:, None] * xt).sum(axis=0) / tot
    s_want = np.sqrt((w * (xt ** 2).sum(axis=1)).sum() / tot.sum())
    assert _rel(new.mean, mean_want) < 1e-12 and abs(new.isotropic_noise - s_want) < 1e-12 * s_want
    assert abs(llk - (w * llks).sum()) < 1e-10 * abs(llk)
    # PPCAModel.init accepts it, the trainer runs on it and the likelihood does not decrease
    m0 = P.PPCAModel.init(0, ds, seed=1)
    assert m0.state_size == 0
    m1 = m0.iterate(ds)
    assert m1.llk(ds) >= m0.llk(ds)
    # a mixture may hold such a component next to ordinary ones (state sizes differ per component, mix.rs:50-71)
    mix = P.PPCAMix([m, P.PPCAModel(0.5, rng.standard_normal((d, 2)), mu + 1.0)], np.zeros(2))
    mix1, mllk = mix.iterate_with_llk(ds)
    assert mix1.state_sizes == [0, 2] and np.isfinite(mllk) and mix1.llk(ds) >= mllk - 1e-9 * abs(mllk)


@pytest.mark.gpu
def test_filter_extrapolate_is_smooth(P):
    """The README's name for the smoothing pass (readme.md:62; the binding calls it `smooth`, src/python_bindings.rs:488-490)."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((300, 24))
    x[rng.random(x.shape) < 0.3] = np.nan
    m = P.PPCAModel(0.4, rng.standard_normal((24, 3)), rng.standard_normal(24))
    ds = P.Dataset(x)
    np.testing.assert_array_equal(m.filter_extrapolate(ds).numpy(), m.smooth(ds).numpy())
    mix = P.PPCAMix([m, P.PPCAModel(0.6, rng.standard_normal((24, 3)), rng.standard_normal(24))], np.log([0.4, 0.6]))
    np.testing.assert_array_equal(mix.filter_extrapolate(ds).numpy(), mix.smooth(ds).numpy())


@pytest.mark.gpu
def test_large_downloads_are_pipelined_and_canonical(P):
    """Dataset.numpy() / infer() of blocks larger than one 64 MB chunk leave through the pipelined copy (canonicalising
    kernel -> pinned chunk buffers -> host threads): masked and +-inf entries come back NaN (dataset.rs:64-72), everything
    else bit for bit; the covariances of a large infer() equal those of its slices."""
    rng = np.random.default_rng(13)
    n, d = 700_000, 32  # 179 MB: three chunks
    x = rng.standard_normal((n, d))
    x[rng.random((n, d)) < 0.2] = np.nan
    x[::1001, 3] = np.inf
    x[::997, 5] = -np.inf
    ds = P.Dataset(x)
    back = ds.numpy()
    want = np.where(np.isfinite(x), x, np.nan)
    assert np.array_equal(np.isnan(back), np.isnan(want))
    np.testing.assert_array_equal(back[np.isfinite(want)], want[np.isfinite(want)])
    m = P.PPCAModel(0.5, rng.standard_normal((d, 4)), rng.standard_normal(d))
    inf = m.infer(ds)  # covariances: 700 000 x 16 doubles = 90 MB
    part = m.infer(ds._slice(650_000, 1000))
    np.testing.assert_array_equal(inf.states()[650_000:651_000], part.states())
    np.testing.assert_array_equal(np.array(inf.covariances()[650_000:651_000]), np.array(part.covariances()))


# --------------------------------------------------------------------------- round 6
def test_multi_component_step_equals_component_by_component(oracle, tmp_path):
    """Round 6: the mixture step launches every stage ONCE over all components (mix_llk8_kernel: the K llk sweeps with X read from
    HBM once; selection, reduction + verdict, finalisation + next slice tables with blockIdx.y = component).  Against the
    component-by-component form of rounds 2-5 (PPCA_MIX_MULTI=0) in a child process each: the llks are BIT-identical (same per-sample
    arithmetic), and so are the new models (same row lists, weights and partial statistics per component); the log-weights and the
    llk total differ only by the order of two sums.  The case with a component outside the int8 Gram's dynamic range must take that
    component's llks from the fp64 instantiation (engine 1) and agree with the oracle."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        path = str(tmp_path / f"mix_multi_{flag}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "mix_multi_check.py"), path],
                           capture_output=True, text=True, env=dict(os.environ, PPCA_MIX_MULTI=flag), timeout=900)
        assert r.returncode == 0 and "mix multi check written" in r.stdout, (flag, r.stdout[-1500:], r.stderr[-2500:])
        outs.append(np.load(path))
    multi, single = outs
    for case in ("k8", "k8_grid8", "k3_grid2", "k3_grid16", "k16"):
        np.testing.assert_array_equal(multi[case + "_llks"], single[case + "_llks"], err_msg=case)
        np.testing.assert_array_equal(multi[case + "_lp"], single[case + "_lp"], err_msg=case)
        np.testing.assert_array_equal(multi[case + "_c"], single[case + "_c"], err_msg=case)
        np.testing.assert_array_equal(multi[case + "_mean"], single[case + "_mean"], err_msg=case)
        np.testing.assert_array_equal(multi[case + "_sigma"], single[case + "_sigma"], err_msg=case)
        for key in ("_llks", "_c", "_mean", "_sigma", "_lw", "_trace"):
            assert np.isfinite(multi[case + key]).all(), (case, key)
        assert np.abs(multi[case + "_lw"] - single[case + "_lw"]).max() < 1e-12, case
        assert _rel(multi[case + "_trace"], single[case + "_trace"]) < 1e-13, case
    np.testing.assert_array_equal(multi["guard_llks"], single["guard_llks"])
    np.testing.assert_array_equal(multi["guard_engine"], np.array([0, 0, 1, 0]))


def test_generic_wp_digits_under_predicted_scales(oracle, tmp_path):
    """Round 6 (split pipeline, ppca_generic.hip): from a call's second chunk on, the int8 digit planes of wP are cut in the pass that
    takes the column statistics, under scales predicted from the chunk before (gen_wdigits_lines_kernel; columns whose maximum left
    the predicted window are cut a second time), and the planes are written as whole lines.  Against the three-kernel form of rounds
    2-5 (PPCA_GEN_WPRED=0), a child process each (tools/wpred_check.py): one chunk -> bit-identical; several chunks -> 1e-11 of each
    other and 1e-8 of the oracle (ppca_model.rs:294-325), including weights that jump by 2^20 between chunks, a chunk of zero
    weights and a chunk whose guard trips."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        path = str(tmp_path / f"wpred_{flag}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "wpred_check.py"), path],
                           capture_output=True, text=True, env=dict(os.environ, PPCA_GEN_WPRED=flag), timeout=900)
        assert r.returncode == 0 and "wpred check written" in r.stdout, (flag, r.stdout[-1500:], r.stderr[-2500:])
        outs.append(np.load(path))
    new, old = outs
    np.testing.assert_array_equal(new["one_chunk"], old["one_chunk"])

    def blocks(d, k):
        kp = k * (k + 1) // 2
        b = [0, d * k, d * k + d * kp, 2 * d * k + d * kp, 2 * d * k + d * kp + d, 2 * d * k + d * kp + 2 * d]
        return list(zip(["cross", "S", "U", "sumx", "totals"], b[:-1], b[1:]))

    def check(name, x, w, sigma, c, mean):
        want = oracle.stats(x, float(sigma), c, mean, w)
        for part, a, b in blocks(*c.shape):
            assert _rel(new[name][a:b], old[name][a:b]) < 1e-11, (name, part)
            assert _rel(new[name][a:b], want[a:b]) < 1e-8, (name, part)
        assert _rel(new[name][b:], want[b:]) < 1e-8, name  # scalars

    x, w = new["in_x"], new["in_w"]
    m = (new["in_sigma"], new["in_c"], new["in_mean"])
    check("one_chunk", x, w, *m)
    check("chunks_256", x, w, *m)
    check("weights_jump", x, new["in_w2"], *m)
    check("zero_chunk", x, new["in_w3"], *m)
    x4 = x.copy()
    x4[700] = new["in_x4_row"]
    check("outlier_chunk", x4, w, *m)
    check("k64", new["k64_x"], None, new["k64_sigma"], new["k64_c"], new["k64_mean"])
    check("d1100", new["d1100_x"], None, new["d1100_sigma"], new["d1100_c"], new["d1100_mean"])


def test_multi_component_sweep_with_a_guard_tripping_component(P, oracle):
    """mix_llk8_kernel skips a component whose slice table tripped the dynamic-range guard; the fp64 instantiation behind the same
    flag serves it (mix.rs:137-149: llks of every component).  Against the oracle, with the tripping component in slot 2 of 4."""
    rng = np.random.default_rng(811)
    d, k, nm, n = 256, 10, 4, 900
    big = np.sort(rng.choice(d, 200, replace=False))
    scale = np.ones(d)
    scale[big] = 1.0e8
    sig = np.array([0.9, 1.1, 0.05, 1.0])
    cs = rng.standard_normal((nm, d, k))
    cs[2] *= scale[:, None]
    ms = 0.1 * rng.standard_normal((nm, d))
    lw = np.log(np.array([0.2, 0.3, 0.1, 0.4]))
    x = rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.3 * rng.standard_normal((n, d))
    x[rng.random((n, d)) < 0.3] = np.nan
    x[:, big] = np.nan
    ds = P.Dataset(x)
    mix = P.PPCAMix([P.PPCAModel(sig[c], cs[c], ms[c]) for c in range(nm)], lw)
    assert [_engine(P, ds, m) for m in mix.models] == [0, 0, 1, 0]
    assert _rel(mix.llks(ds), oracle.mix_llks(x, sig, cs, ms, lw)) < 1e-9
    assert _rel(mix.infer_cluster(ds), oracle.mix_infer_cluster(x, sig, cs, ms, lw)) < 1e-8


@pytest.mark.parametrize("d,k", [(2, 1), (64, 4), (200, 10), (256, 10), (254, 7)])
def test_output_rows_on_the_eight_wave_sweep(P, oracle, d, k):
    """Round 6: PPCAModel.smooth / extrapolate (ppca_model.rs:237-261) for an even d <= 256, k <= 10 run on the eight-wave sweep
    (recon8_kernel: back substitution in the solver step, the rows formed with the staging's lane map and stored as whole 16-byte
    pieces).  Ragged n around the 64-sample round, an all-masked row, a fully observed row, the full grid and ONE workgroup walking
    every round; extrapolate passes the observed values through bit-exactly."""
    from ppca_rs_amd import _lib

    rng = np.random.default_rng(1000 * d + k)
    c, mu, s = rng.standard_normal((d, k)), rng.standard_normal(d), 0.3
    m = P.PPCAModel(s, c, mu)
    ctx = _lib.default_context()
    try:
        for n, limit in ((1, 0), (63, 0), (65, 1), (1000, 0), (1931, 1), (1931, 3)):
            x = rng.standard_normal((n, k)) @ c.T + mu + s * rng.standard_normal((n, d))
            x[rng.random((n, d)) < 0.3] = np.nan
            x[0] = np.nan
            if n > 2:
                x[2] = rng.standard_normal(d)
            ctx.set_grid_limit(limit)
            ds = P.Dataset(x)
            assert _rel(m.smooth(ds).numpy(), oracle.reconstruct(x, s, c, mu, "smooth")) < 1e-9, (n, limit)
            ex = m.extrapolate(ds).numpy()
            assert _rel(ex, oracle.reconstruct(x, s, c, mu, "extrapolate")) < 1e-9, (n, limit)
            assert np.array_equal(ex[np.isfinite(x)], x[np.isfinite(x)]), (n, limit)
            np.testing.assert_allclose(m.smooth(ds).numpy()[0], mu, rtol=0, atol=0)  # the all-masked row: z = 0, the mean exactly
    finally:
        ctx.set_grid_limit(0)
