"""ctypes front-end of the CPU oracle (oracle/ppca_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (ppca_rs_amd) never imports it.

Conventions mirror the reference's Python surface (src/python_bindings.rs):
datasets are float64 (N, d) arrays with non-finite entries = masked
(dataset.rs:19-22), transforms are (d, k), means are (d,).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppca_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ppca_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libppca_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Prior(C.Structure):
    _fields_ = [
        ("has_mean_prior", C.c_int),
        ("mean", _dp),
        ("mean_covariance", _dp),
        ("has_isotropic_noise_prior", C.c_int),
        ("isotropic_noise_alpha", C.c_double),
        ("isotropic_noise_beta", C.c_double),
        ("transformation_precision", C.c_double),
    ]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ppca_oracle_quadratic_form.restype = C.c_double
        _lib.ppca_oracle_covariance_log_det.restype = C.c_double
        _lib.ppca_oracle_llk.restype = C.c_double
        _lib.ppca_oracle_iterate.restype = C.c_int
        _lib.ppca_oracle_mix_iterate.restype = C.c_int
        _lib.ppca_oracle_num_threads.restype = C.c_int
    return _lib


def _a(x, shape=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if shape is not None:
        x = x.reshape(shape)
    return x


def _p(x):
    return x.ctypes.data_as(_dp) if x is not None else None


@dataclass
class Prior:
    """prior.rs:8-65"""
    mean: Optional[np.ndarray] = None
    mean_covariance: Optional[np.ndarray] = None
    isotropic_noise_alpha: Optional[float] = None
    isotropic_noise_beta: Optional[float] = None
    transformation_precision: float = 0.0

    def _c(self):
        keep = []
        p = _Prior()
        p.has_mean_prior = int(self.mean is not None)
        if self.mean is not None:
            m = _a(self.mean).ravel()
            cv = _a(self.mean_covariance)
            keep += [m, cv]
            p.mean, p.mean_covariance = _p(m), _p(cv)
        p.has_isotropic_noise_prior = int(self.isotropic_noise_alpha is not None)
        p.isotropic_noise_alpha = float(self.isotropic_noise_alpha or 0.0)
        p.isotropic_noise_beta = float(self.isotropic_noise_beta or 0.0)
        p.transformation_precision = float(self.transformation_precision)
        return p, keep


def num_threads() -> int:
    return int(lib().ppca_oracle_num_threads())


def set_threads(n: int) -> int:
    """OpenMP threads of the oracle's loops; returns the previous value (tests with very many tiny calls on a many-core host)."""
    return int(lib().ppca_oracle_set_threads(int(n)))


def quadratic_form(sigma, c, x) -> float:
    c = _a(c); x = _a(x).ravel()
    return float(lib().ppca_oracle_quadratic_form(_p(c), c.shape[0], c.shape[1], C.c_double(sigma), _p(x)))


def covariance_log_det(sigma, c) -> float:
    c = _a(c)
    return float(lib().ppca_oracle_covariance_log_det(_p(c), c.shape[0], c.shape[1], C.c_double(sigma)))


def _model(sigma, c, mean):
    c = _a(c)
    mean = _a(mean).ravel()
    assert mean.shape[0] == c.shape[0]
    return float(sigma), c, mean


def llks(x, sigma, c, mean) -> np.ndarray:
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape
    out = np.empty(n)
    lib().ppca_oracle_llks(_p(x), C.c_int64(n), d, c.shape[1], C.c_double(sigma), _p(c), _p(mean), _p(out))
    return out


def llk(x, sigma, c, mean, w=None) -> float:
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape
    w = _a(w).ravel() if w is not None else None
    return float(lib().ppca_oracle_llk(_p(x), _p(w), C.c_int64(n), d, c.shape[1], C.c_double(sigma), _p(c), _p(mean)))


def infer(x, sigma, c, mean):
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape; k = c.shape[1]
    states = np.empty((n, k)); covs = np.empty((n, k, k))
    lib().ppca_oracle_infer(_p(x), C.c_int64(n), d, k, C.c_double(sigma), _p(c), _p(mean), _p(states), _p(covs))
    return states, covs


def reconstruct(x, sigma, c, mean, mode: str) -> np.ndarray:
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape
    out = np.empty((n, d))
    lib().ppca_oracle_reconstruct(_p(x), C.c_int64(n), d, c.shape[1], C.c_double(sigma), _p(c), _p(mean),
                                  {"smooth": 0, "extrapolate": 1}[mode], _p(out))
    return out


def iterate(x, sigma, c, mean, w=None, prior: Optional[Prior] = None):
    """-> (sigma', C', mean').  ppca_model.rs:277-393"""
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape; k = c.shape[1]
    w = _a(w).ravel() if w is not None else None
    c_out = np.empty((d, k)); mean_out = np.empty(d); s_out = C.c_double(0.0)
    pr, keep = (prior._c() if prior is not None else (None, []))
    rc = lib().ppca_oracle_iterate(_p(x), _p(w), C.c_int64(n), d, k, C.c_double(sigma), _p(c), _p(mean),
                                   C.byref(pr) if pr is not None else None, _p(c_out), _p(mean_out), C.byref(s_out))
    if rc != 0:
        raise RuntimeError(f"oracle iterate failed rc={rc}")
    return s_out.value, c_out, mean_out


def stats_len(d, k):
    return 2 * d * k + d * (k * (k + 1) // 2) + 2 * d + 8


def stats(x, sigma, c, mean, w=None) -> np.ndarray:
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape; k = c.shape[1]
    w = _a(w).ravel() if w is not None else None
    out = np.empty(stats_len(d, k))
    lib().ppca_oracle_stats(_p(x), _p(w), C.c_int64(n), d, k, C.c_double(sigma), _p(c), _p(mean), _p(out))
    return out


def fused_stats(x, sigma, c, mean, w=None) -> np.ndarray:
    """The honestly optimised CPU form of the statistics pass (one sweep, Cholesky, OpenMP): a second CPU
    baseline for bench.py, same layout as stats()."""
    x = _a(x); sigma, c, mean = _model(sigma, c, mean)
    n, d = x.shape; k = c.shape[1]
    w = _a(w).ravel() if w is not None else None
    out = np.empty(stats_len(d, k))
    lib().ppca_oracle_fused_stats(_p(x), _p(w), C.c_int64(n), d, k, C.c_double(sigma), _p(c), _p(mean), _p(out))
    return out


def to_canonical(c) -> np.ndarray:
    c = _a(c)
    out = np.empty_like(c)
    if c.shape[1] == 0:
        return c.copy()
    lib().ppca_oracle_to_canonical(_p(c), c.shape[0], c.shape[1], _p(out))
    return out


def singular_values(c) -> np.ndarray:
    """ppca_model.rs:113-121 -- sqrt of the column norm (sic)."""
    return np.sqrt(np.linalg.norm(_a(c), axis=0))


def n_parameters(d, k) -> int:
    """ppca_model.rs:107-109"""
    return 1 + k * d + d


def _mix(sigmas, cs, means, log_weights):
    sigmas = _a(sigmas).ravel(); cs = _a(cs); means = _a(means); lw = _a(log_weights).ravel()
    nm, d, k = cs.shape
    assert means.shape == (nm, d) and sigmas.shape == (nm,) and lw.shape == (nm,)
    return sigmas, cs, means, lw, nm, d, k


def mix_llks(x, sigmas, cs, means, log_weights):
    x = _a(x); sigmas, cs, means, lw, nm, d, k = _mix(sigmas, cs, means, log_weights)
    out = np.empty(x.shape[0])
    lib().ppca_oracle_mix_llks(_p(x), C.c_int64(x.shape[0]), d, k, nm, _p(sigmas), _p(cs), _p(means), _p(lw), _p(out))
    return out


def mix_infer_cluster(x, sigmas, cs, means, log_weights):
    x = _a(x); sigmas, cs, means, lw, nm, d, k = _mix(sigmas, cs, means, log_weights)
    out = np.empty((x.shape[0], nm))
    lib().ppca_oracle_mix_infer_cluster(_p(x), C.c_int64(x.shape[0]), d, k, nm, _p(sigmas), _p(cs), _p(means), _p(lw), _p(out))
    return out


def mix_iterate(x, sigmas, cs, means, log_weights, w=None, prior: Optional[Prior] = None):
    x = _a(x); sigmas, cs, means, lw, nm, d, k = _mix(sigmas, cs, means, log_weights)
    w = _a(w).ravel() if w is not None else None
    s_out = np.empty(nm); c_out = np.empty((nm, d, k)); m_out = np.empty((nm, d)); lw_out = np.empty(nm)
    pr, keep = (prior._c() if prior is not None else (None, []))
    rc = lib().ppca_oracle_mix_iterate(_p(x), _p(w), C.c_int64(x.shape[0]), d, k, nm, _p(sigmas), _p(cs), _p(means), _p(lw),
                                       C.byref(pr) if pr is not None else None, _p(s_out), _p(c_out), _p(m_out), _p(lw_out))
    if rc != 0:
        raise RuntimeError(f"oracle mix_iterate failed rc={rc}")
    return s_out, c_out, m_out, lw_out


def covariance_diagonal(x, sigma, c, mean, mode: str) -> np.ndarray:
    """smoothed_covariance_diagonal ppca_model.rs:485-508 / extrapolated_covariance_diagonal :542-577
    (sigma^2 + <c_j Sigma_i, c_j>; observed dims -> 0 in the extrapolated form), from the oracle's own posteriors."""
    x = _a(x); sigma_, c_, mean_ = _model(sigma, c, mean)
    _, covs = infer(x, sigma, c, mean)
    diag = np.einsum("ja,nab,jb->nj", c_, covs, c_) + float(sigma) ** 2
    return diag if mode == "smooth" else np.where(np.isfinite(x), 0.0, diag)


def mix_inferred(x, sigmas, cs, means, log_weights) -> dict:
    """Literal numpy restatement of InferredMaskedMix's accessors (mix.rs:374-505) on top of the oracle's
    per-component posteriors: state() weights by the LOG posterior (:374-380, as written upstream),
    covariance() :383-396, smoothed :399-407, extrapolated :410-418, the two diagonal covariances :447-461 and
    :485-505 (each around the corresponding mixture mean)."""
    x = _a(x); sigmas, cs, means, lw, nm, d, k = _mix(sigmas, cs, means, log_weights)
    lp = mix_infer_cluster(x, sigmas, cs, means, lw)
    post = np.exp(lp)
    inf = [infer(x, sigmas[c], cs[c], means[c]) for c in range(nm)]
    zs = np.stack([i[0] for i in inf])
    state = np.einsum("nc,cnk->nk", lp, zs)
    cov = np.zeros((x.shape[0], k, k))
    for c in range(nm):
        dv = zs[c] - state
        cov += post[:, c, None, None] * (inf[c][1] + dv[:, :, None] * dv[:, None, :])
    out = {"log_posterior": lp, "state": state, "covariance": cov}
    for mode in ("smooth", "extrapolate"):
        val = [reconstruct(x, sigmas[c], cs[c], means[c], mode) for c in range(nm)]
        mean = np.einsum("nc,cnj->nj", post, np.stack(val))
        dg = [covariance_diagonal(x, sigmas[c], cs[c], means[c], mode) + (val[c] - mean) ** 2 for c in range(nm)]
        out[mode] = mean
        out[mode + "_covariance_diagonal"] = np.einsum("nc,cnj->nj", post, np.stack(dg))
    return out


def synth(n, d, k, mask_prob, seed, sigma_true=0.1, mean_scale=1.0):
    """Seeded generator mirroring sample_one (ppca_model.rs:164-181):
    y = C n1 + mean + sigma n2, entries dropped with probability mask_prob."""
    rng = np.random.default_rng(seed)
    c_true = rng.standard_normal((d, k))
    mean_true = mean_scale * rng.standard_normal(d)
    z = rng.standard_normal((n, k))
    x = z @ c_true.T + mean_true + sigma_true * rng.standard_normal((n, d))
    if mask_prob > 0:
        x[rng.random((n, d)) < mask_prob] = np.nan
    return x, c_true, mean_true
