"""The multi-component form of the mixture step (one launch per stage over all components: mix_llk8_kernel, the select / reduce /
finalise launches with blockIdx.y = component) against the component-by-component form of rounds 2-5.

    python tools/mix_multi_check.py OUT.npz        (PPCA_MIX_MULTI=0 in the environment: the component-by-component form)

Writes the mixture llks / log posteriors of the start mixture and the models, log-weights and llk trace of a few EM iterations for a
list of cases; tests/test_gpu_parity.py::test_multi_component_step_equals_component_by_component runs it twice and compares.  Cases:
K = 8 at d = 256, k = 10 (weighted rows, an all-masked row, a zero weight); K = 3 at d = 40, k = 3 on a grid capped at 2 and at 16
workgroups (several (component, run) units per workgroup); K = 16 at k = 1; one component whose transform trips the dynamic-range
guard of the int8 Gram (rows spanning 1e8, the large ones masked in every sample): its llks come from the fp64 instantiation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ppca_rs_amd as P
from ppca_rs_amd import _lib


def make(rng, d, k, nm, n, mask=0.3, weights=True):
    truth = [(rng.standard_normal((d, k)), 2.5 * rng.standard_normal(d)) for _ in range(nm)]
    comp = rng.integers(0, nm, n)
    x = np.empty((n, d))
    for c, (cc, mu) in enumerate(truth):
        idx = np.nonzero(comp == c)[0]
        x[idx] = rng.standard_normal((len(idx), k)) @ cc.T + mu + 0.2 * rng.standard_normal((len(idx), d))
    x[rng.random((n, d)) < mask] = np.nan
    x[3] = np.nan  # an all-masked row
    w = rng.uniform(0.5, 2.0, n) if weights else None
    if weights:
        w[5] = 0.0
    # (a start near the truth: from random models a component of so small a dataset can collapse onto one row, and every comparison
    #  downstream would be NaN against NaN)
    start = P.PPCAMix([P.PPCAModel(1.0 + 0.1 * c, truth[c][0] + 0.3 * rng.standard_normal((d, k)), truth[c][1] + 0.3 * rng.standard_normal(d))
                       for c in range(nm)], np.log(rng.dirichlet(np.ones(nm) * 3)))
    return x, w, start


def run(name, x, w, start, steps, out, grid_limit=0):
    ctx = _lib.default_context()
    ctx.set_grid_limit(grid_limit)
    ds = P.Dataset(x, w) if w is not None else P.Dataset(x)
    out[name + "_llks"] = start.llks(ds)
    out[name + "_lp"] = start.infer_cluster(ds)
    mix, trace = start, []
    for _ in range(steps):
        mix, llk = mix.iterate_with_llk(ds)
        trace.append(llk)
    out[name + "_trace"] = np.array(trace)
    out[name + "_sigma"] = np.array([m.isotropic_noise for m in mix.models])
    out[name + "_c"] = np.stack([m.transform for m in mix.models])
    out[name + "_mean"] = np.stack([m.mean for m in mix.models])
    out[name + "_lw"] = np.asarray(mix.log_weights)
    ctx.set_grid_limit(0)


def main():
    out = {}
    rng = np.random.default_rng(2024)
    x, w, start = make(rng, 256, 10, 8, 6000)
    run("k8", x, w, start, 3, out)
    run("k8_grid8", x, w, start, 2, out, grid_limit=8)
    x, w, start = make(rng, 40, 3, 3, 3001)
    run("k3_grid2", x, w, start, 3, out, grid_limit=2)
    run("k3_grid16", x, w, start, 2, out, grid_limit=16)
    x, w, start = make(rng, 64, 1, 16, 2000, weights=False)
    run("k16", x, w, start, 2, out)
    # one component outside the int8 Gram's dynamic range (test_int8_gram_dynamic_range_guard (a)): rows of C spanning 1e8, the large
    # ones masked in every sample
    d, k, nm, n = 256, 10, 4, 1500
    x, w, start = make(rng, d, k, nm, n)
    big = np.sort(rng.choice(d, 200, replace=False))
    x[:, big] = np.nan
    models = list(start.models)
    scale = np.ones(d)
    scale[big] = 1.0e8
    models[2] = P.PPCAModel(0.05, rng.standard_normal((d, k)) * scale[:, None], 0.1 * rng.standard_normal(d))
    start = P.PPCAMix(models, start.log_weights)
    ds = P.Dataset(x, w)
    out["guard_llks"] = start.llks(ds)
    out["guard_lp"] = start.infer_cluster(ds)
    out["guard_engine"] = np.array([int(_engine(ds, m)) for m in start.models])
    np.savez(sys.argv[1], **out)
    print("mix multi check written", sys.argv[1])


def _engine(ds, m):
    import ctypes as C

    e = C.c_int32(-1)
    _lib.check(_lib.lib().ppca_gram_engine(ds._ctx.handle, m._device(ds._ctx).h, C.byref(e)))
    return e.value


if __name__ == "__main__":
    main()
