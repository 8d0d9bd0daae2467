"""Seeded inputs of the fixtures whose sample matrices are too large to commit: the fixture holds the expected
outputs and a checksum of the inputs, the inputs are regenerated here (numpy's PCG64 streams only -- no oracle, no
reference code), by tests/golden/make_golden.py when the fixture is written and by the tests when it is read."""
import hashlib

import numpy as np


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def cfg5_inputs(n: int = 2048, d: int = 256, k: int = 10, nm: int = 8, mask_prob: float = 0.3, seed: int = 505):
    """BASELINE config 5 at its own shape (8 components, d = 256, state_size = 10, 30 % masked) with an oracle-sized N:
    samples drawn from 8 different PPCA models, shuffled; non-uniform sample weights; a perturbed start mixture with
    unequal weights.  Returns x, w, sigmas, cs, means, log_weights."""
    rng = np.random.default_rng(seed)
    parts = []
    truth_c = 0.7 * rng.standard_normal((nm, d, k))
    truth_m = 0.5 * rng.standard_normal((nm, d))
    per = n // nm
    for c in range(nm):
        z = rng.standard_normal((per, k))
        parts.append(z @ truth_c[c].T + truth_m[c] + 0.3 * rng.standard_normal((per, d)))
    x = np.concatenate(parts)
    rng.shuffle(x)
    x[rng.random(x.shape) < mask_prob] = np.nan
    x[5, :] = np.nan  # an all-masked sample
    w = rng.uniform(0.25, 2.0, x.shape[0])
    # start: small perturbations of ONE model, so that the first responsibilities are soft (at d = 256 any two
    # well-separated components give posteriors of exactly 0 / 1) and sharpen over the iterations
    base_c, base_m = truth_c.mean(axis=0), truth_m.mean(axis=0)
    cs = base_c + 0.04 * rng.standard_normal((nm, d, k))
    ms = base_m + 0.04 * rng.standard_normal((nm, d))
    sig = rng.uniform(0.9, 1.1, nm)
    lw = np.log(rng.dirichlet(2.0 * np.ones(nm)))
    return x, w, sig, cs, ms, lw
