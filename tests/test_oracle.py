"""Pins the CPU oracle: reference KATs, an independent dense-Gaussian evaluation,
documented invariants, and the committed golden vectors (CPU only)."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
              if not os.path.basename(p).startswith(("mix_", "cfg5_")))
GOLD_MIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mix_*.npz")))
C_TOY = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])  # ppca_model.rs:647-656


def test_kat_quadratic_form(oracle):
    # ppca_model.rs:658-665 asserts 34.219288 (a 6-digit literal); exact value 34.219269102989976
    v = oracle.quadratic_form(0.1, C_TOY, [1.0, 1.0, 1.0])
    assert abs(v - 34.219288) < 5e-5
    assert abs(v - 34.219269102989976) < 1e-10


def test_kat_covariance_log_det(oracle):
    # ppca_model.rs:667-671 asserts -3.49328
    v = oracle.covariance_log_det(0.1, C_TOY)
    assert abs(v - (-3.49328)) < 5e-6
    assert abs(v - (-3.4932797763741386)) < 1e-12


def test_llk_toy_model_dense_gaussian(oracle):
    # ppca_model.rs:673-680: inputs only upstream; expected from a dense N(mean, C C^T + s^2 I)
    from scipy.stats import multivariate_normal as mvn

    x = np.array([[1.0, 2.0, 3.0]])
    mean = np.array([0.0, 1.0, 0.0])
    want = mvn.logpdf(x[0], mean, C_TOY @ C_TOY.T + 0.01 * np.eye(3))
    assert abs(oracle.llk(x, 0.1, C_TOY, mean) - want) < 1e-9
    assert abs(want - (-152.9969524621925)) < 1e-8


def _dense_check(oracle, x, s, c, mu):
    from scipy.stats import multivariate_normal as mvn

    l = oracle.llks(x, s, c, mu)
    st, cv = oracle.infer(x, s, c, mu)
    ex = oracle.reconstruct(x, s, c, mu, "extrapolate")
    k = c.shape[1]
    for i in range(len(x)):
        ob = np.isfinite(x[i])
        if ob.sum() == 0:
            assert l[i] == 0.0 and np.all(st[i] == 0) and np.array_equal(cv[i], np.eye(k))
            continue
        co = c[ob]
        cov = co @ co.T + s * s * np.eye(ob.sum())
        assert abs(l[i] - mvn.logpdf(x[i, ob], mu[ob], cov)) < 1e-8 * max(1.0, abs(l[i]))
        m = co.T @ co + s * s * np.eye(k)
        z = np.linalg.solve(m, co.T @ (x[i, ob] - mu[ob]))
        np.testing.assert_allclose(st[i], z, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(cv[i], s * s * np.linalg.inv(m), rtol=1e-7, atol=1e-10)
        # conditional mean of the missing dims given the observed ones
        if (~ob).any():
            cm = mu[~ob] + c[~ob] @ co.T @ np.linalg.solve(cov, x[i, ob] - mu[ob])
            np.testing.assert_allclose(ex[i, ~ob], cm, rtol=1e-7, atol=1e-9)
        np.testing.assert_array_equal(ex[i, ob], x[i, ob])


def test_dense_gaussian_agreement(oracle):
    x, _, _ = oracle.synth(120, 10, 3, 0.35, 5)
    x[7] = np.nan
    rng = np.random.default_rng(3)
    _dense_check(oracle, x, 0.6, rng.standard_normal((10, 3)), 0.2 * rng.standard_normal(10))


def test_em_monotone_and_canonical_invariance(oracle):
    # ppca_model.rs:263-265 (llk never decreases), :395-397 (canonical form keeps the llk)
    x, _, _ = oracle.synth(400, 9, 2, 0.2, 9)
    rng = np.random.default_rng(4)
    s, c, mu = 1.0, rng.standard_normal((9, 2)), np.zeros(9)
    prev = -np.inf
    for _ in range(12):
        l = oracle.llk(x, s, c, mu)
        assert l >= prev - 1e-8 * abs(l)
        prev = l
        s, c, mu = oracle.iterate(x, s, c, mu)
    cn = oracle.to_canonical(c)
    assert abs(oracle.llk(x, s, cn, mu) - oracle.llk(x, s, c, mu)) < 1e-8 * abs(prev)
    u, sv, _ = np.linalg.svd(c, full_matrices=False)
    ref = u * sv
    ref = ref * np.where(np.signbit(ref.sum(0)), -1.0, 1.0)
    np.testing.assert_allclose(cn, ref, atol=1e-10)


def test_stats_reproduce_iterate(oracle):
    """The packed statistics (include/ppca_hip.h layout) carry everything iterate needs."""
    x, _, _ = oracle.synth(150, 7, 3, 0.3, 13)
    rng = np.random.default_rng(8)
    s, c, mu = 0.9, rng.standard_normal((7, 3)), 0.1 * rng.standard_normal(7)
    w = rng.uniform(0.2, 3.0, 150)
    st = oracle.stats(x, s, c, mu, w)
    d, k, kp = 7, 3, 6
    cross = st[: d * k].reshape(d, k)
    S = st[d * k: d * k + d * kp].reshape(d, kp)
    U = st[d * k + d * kp: 2 * d * k + d * kp].reshape(d, k)
    sumx = st[2 * d * k + d * kp: 2 * d * k + d * kp + d]
    tot = st[2 * d * k + d * kp + d: 2 * d * k + d * kp + 2 * d]
    sc = st[-8:]
    s1, c1, m1 = oracle.iterate(x, s, c, mu, w)
    tril = np.tril_indices(k)
    for j in range(d):
        sm = np.zeros((k, k))
        sm[tril] = S[j]
        sm = sm + sm.T - np.diag(np.diag(sm))
        np.testing.assert_allclose(np.linalg.solve(sm, cross[j]), c1[j], rtol=1e-8)
    np.testing.assert_allclose(np.sqrt((sc[0] + sc[1]) / tot.sum()), s1, rtol=1e-10)
    np.testing.assert_allclose((sumx - (c * U).sum(1)) / tot + mu, m1, rtol=1e-9, atol=1e-12)
    assert abs(sc[2] - oracle.llk(x, s, c, mu, w)) < 1e-9 * abs(sc[2])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_golden_vectors(oracle, path):
    g = np.load(path)
    x, s, c, mu = g["x"], float(g["s0"]), g["c0"], g["mu0"]
    w = g["w"] if "w" in g else None
    np.testing.assert_allclose(oracle.llks(x, s, c, mu), g["llks"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(oracle.llk(x, s, c, mu, w), g["llk"], rtol=1e-12)
    st, cv = oracle.infer(x, s, c, mu)
    np.testing.assert_allclose(st, g["states"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(cv, g["covs"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(oracle.reconstruct(x, s, c, mu, "extrapolate"), g["extrapolate"], rtol=1e-10, atol=1e-12)


def test_mixture_oracle_consistency(oracle):
    x, _, _ = oracle.synth(200, 6, 2, 0.2, 17)
    rng = np.random.default_rng(5)
    nm = 3
    sig = np.array([0.7, 1.1, 0.9])
    cs = rng.standard_normal((nm, 6, 2))
    ms = rng.standard_normal((nm, 6))
    lw = np.log(np.array([0.2, 0.5, 0.3]))
    lp = oracle.mix_infer_cluster(x, sig, cs, ms, lw)
    np.testing.assert_allclose(np.exp(lp).sum(1), 1.0, rtol=1e-12)
    l = oracle.mix_llks(x, sig, cs, ms, lw)
    comp = np.stack([oracle.llks(x, sig[c], cs[c], ms[c]) + lw[c] for c in range(nm)], 1)
    np.testing.assert_allclose(l, np.log(np.exp(comp - comp.max(1, keepdims=True)).sum(1)) + comp.max(1), rtol=1e-12)
    prev = l.sum()
    for _ in range(5):  # mix.rs:267-270: llk never decreases
        sig, cs, ms, lw = oracle.mix_iterate(x, sig, cs, ms, lw)
        cur = oracle.mix_llks(x, sig, cs, ms, lw).sum()
        assert cur >= prev - 1e-8 * abs(cur)
        prev = cur


def test_fused_cpu_stats_match_literal_stats(oracle):
    o = oracle
    """bench.py's second CPU baseline (one-sweep Cholesky form) against the literal restatement's statistics."""
    x, _, _ = o.synth(600, 40, 6, 0.35, 77)
    rng = np.random.default_rng(5)
    c0, mu0, w = rng.standard_normal((40, 6)), rng.standard_normal(40) * 0.1, rng.uniform(0.2, 2.0, 600)
    x[3] = np.nan  # an all-masked sample
    for weights in (None, w):
        want = o.stats(x, 0.7, c0, mu0, weights)
        got = o.fused_stats(x, 0.7, c0, mu0, weights)
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("path", GOLD_MIX, ids=[os.path.basename(p) for p in GOLD_MIX])
def test_golden_mixture_vectors(oracle, path):
    """The committed mixture fixture is what the oracle computes today (tests/golden/make_golden.py::mix_case)."""
    g = np.load(path)
    x, w, sig, cs, ms, lw = g["x"], g["w"], g["sig0"], g["cs0"], g["ms0"], g["lw0"]
    np.testing.assert_allclose(oracle.mix_llks(x, sig, cs, ms, lw), g["llks"], rtol=1e-12)
    inf = oracle.mix_inferred(x, sig, cs, ms, lw)
    for key in ("log_posterior", "state", "covariance", "smooth", "extrapolate", "smooth_covariance_diagonal",
                "extrapolate_covariance_diagonal"):
        np.testing.assert_allclose(inf[key], g["inf_" + key], rtol=1e-10, atol=1e-12)
    for it in range(len(g["it_llk"])):
        assert abs(float((oracle.mix_llks(x, sig, cs, ms, lw) * w).sum()) - g["it_llk"][it]) < 1e-10 * abs(g["it_llk"][it])
        sig, cs, ms, lw = oracle.mix_iterate(x, sig, cs, ms, lw, w)
        np.testing.assert_allclose(sig, g["it_sigma"][it], rtol=1e-10)
        np.testing.assert_allclose(cs, g["it_c"][it], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lw, g["it_lw"][it], rtol=1e-10, atol=1e-13)


def _cfg5():
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from inputs import cfg5_inputs, digest

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg5_d256_k10_m8.npz"))
    inp = cfg5_inputs()
    assert digest(*inp) == str(g["digest"]), "tests/golden/inputs.py no longer regenerates the fixture's inputs"
    return g, inp


def test_config5_fixture_is_what_the_oracle_computes(oracle):
    """BASELINE config 5 at its own shape (K = 8, d = 256, k = 10, weighted): the committed fixture (inputs
    regenerated from a seed, checksum checked) against the oracle, tests/golden/make_golden.py::cfg5_case."""
    g, (x, w, sig, cs, ms, lw) = _cfg5()
    np.testing.assert_allclose(oracle.mix_llks(x, sig, cs, ms, lw), g["llks"], rtol=1e-12)
    np.testing.assert_allclose(oracle.mix_infer_cluster(x, sig, cs, ms, lw), g["log_posterior"], rtol=1e-10, atol=1e-10)
    for it in range(len(g["it_llk"])):
        assert abs(float((oracle.mix_llks(x, sig, cs, ms, lw) * w).sum()) - g["it_llk"][it]) < 1e-10 * abs(g["it_llk"][it])
        sig, cs, ms, lw = oracle.mix_iterate(x, sig, cs, ms, lw, w)
        np.testing.assert_allclose(sig, g["it_sigma"][it], rtol=1e-10)
        np.testing.assert_allclose(cs, g["it_c"][it], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lw, g["it_lw"][it], rtol=1e-10, atol=1e-13)


# --------------------------------------------------------------------------- second restatement (round 2)
def _restate():
    from oracle import restate_numpy

    return restate_numpy


def _case(seed, n=70, d=9, k=3, mask=0.3):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.3 * rng.standard_normal((n, d)) + rng.standard_normal(d)
    x[rng.random((n, d)) < mask] = np.nan
    x[4, :] = np.nan   # an all-masked sample
    x[:, 5] = np.nan   # an empty dimension: its row of C and its mean are kept
    return x, rng.uniform(0.3, 2.0, n), 0.8, rng.standard_normal((d, k)), 0.2 * rng.standard_normal(d), rng


def test_closed_form_em_step(oracle):
    """A hand-derivable EM step (ppca_model.rs:277-393): d = 2, k = 1, C = (1, 2)^T, sigma = 1, mean = 0, two fully
    observed samples x1 = (1, 0), x2 = (0, 3).  M = 6, z = (1/6, 1), Sigma = 1/6, P = (7/36, 7/6), S = 49/36,
    cross = (1/6, 3)  =>  C' = (6/49, 108/49); deviations (5/6, -1/3), (-1, 1), square_error 2 * 5/6
    =>  sigma'^2 = (5/3 + 29/36 + 2) / 4 = 161/144; mean' = (-1/12, 1/3).  Both restatements must hit the rationals."""
    x = np.array([[1.0, 0.0], [0.0, 3.0]])
    c, mu = np.array([[1.0], [2.0]]), np.zeros(2)
    want = (np.sqrt(161.0) / 12.0, np.array([[6.0 / 49.0], [108.0 / 49.0]]), np.array([-1.0 / 12.0, 1.0 / 3.0]))
    for got in (oracle.iterate(x, 1.0, c, mu), _restate().iterate_with_prior(x, 1.0, c, mu)):
        assert abs(got[0] - want[0]) < 1e-15
        np.testing.assert_allclose(got[1], want[1], rtol=1e-14)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-14)
    z, cov = oracle.infer(x, 1.0, c, mu)
    np.testing.assert_allclose(z[:, 0], [1.0 / 6.0, 1.0], rtol=1e-14)
    np.testing.assert_allclose(cov[:, 0, 0], [1.0 / 6.0, 1.0 / 6.0], rtol=1e-13)


@pytest.mark.parametrize("which", ["none", "ridge", "noise", "mean", "all"])
def test_second_restatement_agrees_on_the_em_step(oracle, which):
    """oracle/ppca_oracle.c against oracle/restate_numpy.py (written independently from the Rust) on iterate_with_prior
    with each prior hook (ppca_model.rs:307-308, :360-371, :379-384; prior.rs:97-110), weighted, with an all-masked
    sample and an empty dimension: 1e-12."""
    R = _restate()
    x, w, s, c, mu, rng = _case(100 + len(which))
    d = x.shape[1]
    a = rng.standard_normal((d, d))
    cov = a @ a.T / d + 0.5 * np.eye(d)
    kw = {}
    if which in ("ridge", "all"):
        kw["transformation_precision"] = 0.7
    if which in ("noise", "all"):
        kw.update(isotropic_noise_alpha=3.0, isotropic_noise_beta=2.0)
    if which in ("mean", "all"):
        kw.update(mean=np.linspace(-1, 1, d), mean_covariance=cov)
    po = oracle.Prior(**kw) if kw else None
    pn = R.PriorN(**kw) if kw else None
    np.testing.assert_allclose(oracle.llks(x, s, c, mu), R.llks(x, s, c, mu), rtol=1e-12, atol=1e-12)
    for _ in range(2):
        got = oracle.iterate(x, s, c, mu, w, po)
        want = R.iterate_with_prior(x, s, c, mu, w, pn)
        assert abs(got[0] - want[0]) < 1e-12 * want[0]
        np.testing.assert_allclose(got[1], want[1], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-10, atol=1e-12)
        if "transformation_precision" not in kw:
            np.testing.assert_array_equal(got[1][5], c[5])  # empty dimension: singular system, old row kept (:313-321)
        else:
            np.testing.assert_array_equal(got[1][5], np.zeros(c.shape[1]))  # tau I is solvable: the row shrinks to 0
        s, c, mu = got


def test_second_restatement_agrees_on_the_mixture_step(oracle):
    """mix.rs:281-337 in both restatements, weighted, three components."""
    R = _restate()
    x, w, _, _, _, rng = _case(7, n=90, d=8, k=2, mask=0.25)
    nm = 3
    sig, cs, ms = rng.uniform(0.6, 1.2, nm), rng.standard_normal((nm, 8, 2)), rng.standard_normal((nm, 8))
    lw = np.log(rng.dirichlet(np.ones(nm)))
    got = oracle.mix_iterate(x, sig, cs, ms, lw, w)
    want = R.mix_iterate(x, sig, list(cs), list(ms), lw, w)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-11)
    np.testing.assert_allclose(got[1], np.array(want[1]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(got[2], np.array(want[2]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(got[3], want[3], rtol=1e-11, atol=1e-12)
