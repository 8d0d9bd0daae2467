"""Generates tests/golden/*.npz with the CPU oracle (oracle/ppca_oracle.c).

The reference (Rust) cannot be built or imported in this image, so these vectors
are produced by the oracle, which is itself pinned to the reference's two KATs
and to an independent dense-Gaussian evaluation (tests/test_oracle.py).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ppca_oracle as o  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, n, d, k, mask_prob, seed, iters=3, weights=False, prior=None):
    x, _, _ = o.synth(n, d, k, mask_prob, seed)
    if n > 5:
        x[3, :] = np.nan  # an all-masked sample
    rng = np.random.default_rng(seed + 1000)
    c0 = rng.standard_normal((d, k))
    mu0 = 0.1 * rng.standard_normal(d)
    s0 = 0.8
    w = rng.uniform(0.5, 2.0, n) if weights else None
    out = dict(x=x, c0=c0, mu0=mu0, s0=np.float64(s0))
    if w is not None:
        out["w"] = w
    out["llks"] = o.llks(x, s0, c0, mu0)
    out["llk"] = np.float64(o.llk(x, s0, c0, mu0, w))
    st, cv = o.infer(x, s0, c0, mu0)
    out["states"], out["covs"] = st, cv
    out["smooth"] = o.reconstruct(x, s0, c0, mu0, "smooth")
    out["extrapolate"] = o.reconstruct(x, s0, c0, mu0, "extrapolate")
    out["stats"] = o.stats(x, s0, c0, mu0, w)
    sig, c, mu = s0, c0, mu0
    sigs, cs, mus, llks = [], [], [], []
    for _ in range(iters):
        llks.append(o.llk(x, sig, c, mu, w))
        sig, c, mu = o.iterate(x, sig, c, mu, w, prior)
        sigs.append(sig); cs.append(c); mus.append(mu)
    out["it_sigma"], out["it_c"], out["it_mean"], out["it_llk"] = np.array(sigs), np.array(cs), np.array(mus), np.array(llks)
    out["canonical"] = o.to_canonical(c)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "llk", out["llk"], "sigma", sigs)


def mix_case(name, n, d, k, nm, seed, iters=3):
    """A mixture fixture (in its own directory entry `mix_*.npz`): inputs, three iterations of the oracle's mixture
    EM with weights, and the mixture's inference outputs (mix.rs:137-189, :281-337, :352-505)."""
    rng = np.random.default_rng(seed)
    x = np.concatenate([o.synth(n // nm, d, k, 0.3, seed + 10 * c, mean_scale=2.5)[0] for c in range(nm)])
    rng.shuffle(x)
    x[2, :] = np.nan
    w = rng.uniform(0.5, 2.0, x.shape[0])
    sig, cs, ms = rng.uniform(0.6, 1.2, nm), rng.standard_normal((nm, d, k)), rng.standard_normal((nm, d))
    lw = np.log(rng.dirichlet(np.ones(nm)))
    out = dict(x=x, w=w, sig0=sig, cs0=cs, ms0=ms, lw0=lw)
    out["llks"] = o.mix_llks(x, sig, cs, ms, lw)
    inf = o.mix_inferred(x, sig, cs, ms, lw)
    for key in ("log_posterior", "state", "covariance", "smooth", "extrapolate", "smooth_covariance_diagonal",
                "extrapolate_covariance_diagonal"):
        out["inf_" + key] = inf[key]
    sigs, css, mss, lws, llks = [], [], [], [], []
    for _ in range(iters):
        llks.append(float((o.mix_llks(x, sig, cs, ms, lw) * w).sum()))
        sig, cs, ms, lw = o.mix_iterate(x, sig, cs, ms, lw, w)
        sigs.append(sig); css.append(cs); mss.append(ms); lws.append(lw)
    out["it_sigma"], out["it_c"], out["it_mean"], out["it_lw"], out["it_llk"] = map(np.array, (sigs, css, mss, lws, llks))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "llk", llks)


def cfg5_case(name, iters=2):
    """BASELINE config 5 at its own shape (K = 8, d = 256, k = 10, 30 % masked, weighted), oracle-sized N.  The
    inputs are regenerated from a seed (tests/golden/inputs.py); the fixture holds their checksum and the oracle's
    outputs: per-sample mixture llks, log posteriors, and `iters` weighted mixture EM iterations (mix.rs:137-189,
    :281-337)."""
    from inputs import cfg5_inputs, digest

    x, w, sig, cs, ms, lw = cfg5_inputs()
    out = dict(digest=np.array(digest(x, w, sig, cs, ms, lw)), llks=o.mix_llks(x, sig, cs, ms, lw),
               log_posterior=o.mix_infer_cluster(x, sig, cs, ms, lw))
    sigs, css, mss, lws, llks = [], [], [], [], []
    for _ in range(iters):
        llks.append(float((o.mix_llks(x, sig, cs, ms, lw) * w).sum()))
        sig, cs, ms, lw = o.mix_iterate(x, sig, cs, ms, lw, w)
        sigs.append(sig); css.append(cs); mss.append(ms); lws.append(lw)
    out["it_sigma"], out["it_c"], out["it_mean"], out["it_lw"], out["it_llk"] = map(np.array, (sigs, css, mss, lws, llks))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "llk", llks)


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    if "--only-cfg5" in sys.argv:
        cfg5_case("cfg5_d256_k10_m8")
        sys.exit(0)
    cfg5_case("cfg5_d256_k10_m8")
    mix_case("mix_d16_k3_m3", 360, 16, 3, 3, 71)
    case("toy_d3_k2", 100, 3, 2, 0.2, 11, iters=5)
    case("small_d12_k3", 300, 12, 3, 0.3, 21, iters=3)
    case("weighted_d20_k4", 200, 20, 4, 0.3, 31, iters=3, weights=True)
    case("dense_d32_k4", 256, 32, 4, 0.0, 41, iters=3)
    case("wide_d256_k10", 96, 256, 10, 0.3, 51, iters=2)
    case("prior_d8_k2", 120, 8, 2, 0.25, 61, iters=3,
         prior=o.Prior(mean=np.linspace(-1, 1, 8), mean_covariance=0.5 * np.eye(8) + 0.1,
                       isotropic_noise_alpha=3.0, isotropic_noise_beta=2.0, transformation_precision=0.7))
