"""TEST INFRASTRUCTURE -- a SECOND, independently written restatement of the reference's EM step in plain numpy.

oracle/ppca_oracle.c is one transcription of viodotcom/ppca_rs; the reference itself pins almost nothing of it
(two 2x2 helper values, tests/test_oracle.py).  This file restates the same functions again, straight from the Rust
source and without looking at the C, one sample at a time with numpy.linalg doing what nalgebra does
(try_inverse -> inv, determinant -> det, qr().solve -> qr + back substitution with the None-on-zero-pivot rule).
tests/test_oracle.py asserts that the two restatements agree to 1e-12 on seeded inputs, for all three prior hooks
and for the mixture step.  Pure-Python loops: small cases only.  Only tests/ may import this module.

Reference (paths relative to the reference root):
  OutputCovariance      ppca/src/output_covariance.rs:57-142
  llk_one / infer_one   ppca/src/ppca_model.rs:124-139, :195-208
  iterate_with_prior    ppca/src/ppca_model.rs:277-393
  Prior::smooth_mean    ppca/src/prior.rs:97-110
  PPCAMix::iterate_...  ppca/src/mix.rs:14-18, :281-337
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

LN_2PI = math.log(2.0 * math.pi)  # ppca_model.rs:16


@dataclass
class PriorN:
    """prior.rs:8-15"""
    mean: Optional[np.ndarray] = None
    mean_covariance: Optional[np.ndarray] = None
    isotropic_noise_alpha: Optional[float] = None
    isotropic_noise_beta: Optional[float] = None
    transformation_precision: float = 0.0


# ---- OutputCovariance (output_covariance.rs) on a sub-transform (the rows of C a mask keeps, :124-131)
def _inner_matrix(sigma, ct):  # :61-64
    k = ct.shape[1]
    return np.eye(k) * sigma ** 2 + ct.T @ ct


def _estimator_transform(sigma, ct):  # :86-90  (the subtractive Woodbury form, as written)
    return (ct.T - (ct.T @ ct) @ np.linalg.inv(_inner_matrix(sigma, ct)) @ ct.T) / sigma ** 2


def _estimator_covariance(sigma, ct):  # :94-97
    return np.eye(ct.shape[1]) - _estimator_transform(sigma, ct) @ ct


def _covariance_log_det(sigma, ct):  # :115-121
    return math.log(np.linalg.det(_inner_matrix(sigma, ct))) + math.log(sigma) * 2.0 * (ct.shape[0] - ct.shape[1])


def _quadratic_form(sigma, ct, x):  # :133-142
    t = ct.T @ x
    return (x @ x - t @ np.linalg.inv(_inner_matrix(sigma, ct)) @ t) / sigma ** 2


def llk_one(sigma, c, mean, row):  # ppca_model.rs:124-139
    obs = np.isfinite(row)
    if not obs.any():
        return 0.0
    sub = (row - mean)[obs]
    ct = c[obs]
    return -_quadratic_form(sigma, ct, sub) / 2.0 - _covariance_log_det(sigma, ct) / 2.0 - LN_2PI / 2.0 * ct.shape[0]


def llks(x, sigma, c, mean):
    return np.array([llk_one(sigma, c, mean, r) for r in x])


def infer_one(sigma, c, mean, row):  # :195-208, uninferred :98-104
    k = c.shape[1]
    obs = np.isfinite(row)
    if not obs.any():
        return np.zeros(k), np.eye(k)
    ct = c[obs]
    return _estimator_transform(sigma, ct) @ (row - mean)[obs], _estimator_covariance(sigma, ct)


def _qr_solve(a, b):
    """nalgebra qr().solve: None when R has a zero on its diagonal."""
    q, r = np.linalg.qr(a)
    if np.any(np.diag(r) == 0.0):
        return None
    y = q.T @ b
    out = np.zeros_like(y)
    for i in reversed(range(len(y))):
        out[i] = (y[i] - r[i, i + 1:] @ out[i + 1:]) / r[i, i]
    return out


def iterate_with_prior(x, sigma, c, mean, w=None, prior: Optional[PriorN] = None):
    """ppca_model.rs:277-393.  Returns (sigma', C', mean')."""
    prior = prior or PriorN()
    n, d = x.shape
    k = c.shape[1]
    w = np.ones(n) if w is None else np.asarray(w, dtype=np.float64)
    obs = np.isfinite(x)
    inferred = [infer_one(sigma, c, mean, x[i]) for i in range(n)]  # :278
    # :281-293  total cross moment (fillna: masked entries contribute 0)
    cross = np.zeros((d, k))
    for i in range(n):
        filled = np.where(obs[i], x[i] - mean, 0.0)
        cross += w[i] * np.outer(filled, inferred[i][0])
    # :294-325  one k x k system per output dimension
    new_c = np.zeros((d, k))
    for j in range(d):
        second = np.zeros((k, k))
        for i in range(n):
            if obs[i, j]:
                z, cov = inferred[i]
                second += w[i] * (np.outer(z, z) + cov)  # second_moment :437-439
        second += prior.transformation_precision * np.eye(k)
        sol = _qr_solve(second, cross[j])
        new_c[j] = sol if sol is not None else c[j]
    # :328-358  noise 4-tuple over the non-empty samples
    sq_err = 0.0
    dev_sq = 0.0
    tot_dev = np.zeros(d)
    totals = np.zeros(d)
    any_sample = False
    for i in range(n):
        if not obs[i].any():
            continue
        any_sample = True
        z, cov = inferred[i]
        ct = c[obs[i]]
        dev = np.where(obs[i], x[i] - c @ z - mean, 0.0)
        sq_err += w[i] * np.sum((ct @ cov) * ct)  # (sub_transform * covariance).dot(sub_transform)
        dev_sq += w[i] * (dev @ dev)
        tot_dev += w[i] * dev
        totals += w[i] * obs[i].astype(np.float64)
    if not any_sample:
        raise ValueError("non-empty dataset")  # :358
    # :360-371
    if prior.isotropic_noise_alpha is not None:
        s2 = ((sq_err + dev_sq) / 2.0 + prior.isotropic_noise_beta) / (totals.sum() / 2.0 + prior.isotropic_noise_alpha + 1.0)
    else:
        s2 = (sq_err + dev_sq) / totals.sum()
    # :373-377
    new_mean = np.where(totals > 0.0, tot_dev / np.where(totals > 0.0, totals, 1.0), 0.0) + mean
    # :379-384 with prior.rs:97-110
    if prior.mean is not None:
        prior_precision = np.linalg.inv(prior.mean_covariance)  # prior.rs:36-41
        precision = np.diag(totals) / s2
        new_mean = _qr_solve(prior_precision + precision, prior_precision @ prior.mean + precision @ new_mean)
    return math.sqrt(s2), new_c, new_mean


def _robust_log_softmax(v):  # mix.rs:14-18
    mx = v.max()
    return v - mx - math.log(np.exp(v - mx).sum())


def mix_iterate(x, sigmas, cs, means, log_weights, w=None, prior: Optional[PriorN] = None):
    """mix.rs:281-337.  cs / means: per-component arrays (components may differ in state size).  Returns
    (sigmas', cs', means', log_weights')."""
    n = x.shape[0]
    w = np.ones(n) if w is None else np.asarray(w, dtype=np.float64)
    nm = len(cs)
    comp_llks = [llks(x, sigmas[m], cs[m], means[m]) for m in range(nm)]  # :283-288
    log_post = np.array([_robust_log_softmax(np.array([comp_llks[m][i] for m in range(nm)]) + log_weights)
                         for i in range(n)])  # :289-295
    out_s, out_c, out_m, logsums = [], [], [], []
    for m in range(nm):
        keep = w > 0.0  # :304-309 (the reference then pairs these n' weights with ALL n samples, :326:
        if not keep.all():  # a length mismatch it never checks; only w > 0 is restated)
            raise ValueError("restated for positive weights only")
        lp = np.log(w) + log_post[:, m]
        mx = np.nanmax(lp)  # :312-317 (NaNs skipped)
        unnorm = np.exp(lp - mx)  # :320-323
        logsums.append(math.log(unnorm.sum()) + mx)  # :324-325
        s1, c1, m1 = iterate_with_prior(x, sigmas[m], cs[m], means[m], unnorm, prior)  # :326-328
        out_s.append(s1); out_c.append(c1); out_m.append(m1)
    return np.array(out_s), out_c, out_m, _robust_log_softmax(np.array(logsums))  # :335
