"""The split pipeline's statistics (ppca_generic.hip) with the wP digit planes cut in the statistics pass under predicted scales
(round 6: gen_wdigits_lines_kernel + the per-column second cut) against the three-kernel form of rounds 2-5 (PPCA_GEN_WPRED=0).

    python tools/wpred_check.py OUT.npz        (PPCA_GEN_WPRED=0 in the environment: statistics, scales, digits as in round 5)

Writes ppca_stats_raw of a list of cases (name -> statistics vector); tests/test_gpu_parity.py::test_generic_wp_digits_under_predicted_scales
runs it once per form and compares: one-chunk cases must be BIT-identical (same scales; the line layout and the 32-bit digit arithmetic
are the only differences), several-chunk cases agree to 1e-11 and with the oracle.  Cases: a single chunk; chunks of 256 rows; the same
with weights that jump by 2^20 between chunks (every column of the later chunk is cut a second time); a chunk whose rows all carry
weight zero (columns of exact zeros) between ordinary ones; a chunk with one row 1e6 x the others (its guard sends it to the fp64
product); k = 64 at d = 300 and k = 20 at d = 1100 (256-row tiles of the statistics product)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ppca_rs_amd as P
from ppca_rs_amd import _lib


def data(rng, n, d, k, mask=0.35):
    x = rng.standard_normal((n, k)) @ rng.standard_normal((k, d)) + 0.2 * rng.standard_normal((n, d)) + rng.standard_normal(d)
    x[rng.random((n, d)) < mask] = np.nan
    x[min(7, n - 1)] = np.nan  # an all-masked row
    m = (0.4 + rng.random(), 0.5 * rng.standard_normal((d, k)), rng.standard_normal(d))
    return x, m


def stats(x, w, m, chunk):
    if chunk:
        os.environ["PPCA_GEN_CHUNK"] = str(chunk)
    else:
        os.environ.pop("PPCA_GEN_CHUNK", None)
    d, k = m[1].shape
    assert _lib.lib().ppca_path_kind(d, k) == 0
    ds, mod = P.Dataset(x, w), P.PPCAModel(*m)
    out = np.empty(_lib.lib().ppca_stats_len(d, k))
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, mod._device(ds._ctx).h, _lib.ptr(out)))
    again = np.empty_like(out)
    _lib.check(_lib.lib().ppca_stats_raw(ds._ctx.handle, ds._h, mod._device(ds._ctx).h, _lib.ptr(again)))
    assert np.array_equal(out, again), "not bit-reproducible"
    return out


def main():
    rng = np.random.default_rng(606)
    out = {}
    x, m = data(rng, 1500, 300, 24)
    w = rng.uniform(0.5, 2.0, len(x))
    out["one_chunk"] = stats(x, w, m, 0)
    out["chunks_256"] = stats(x, w, m, 256)
    w2 = w.copy()
    w2[512:1024] *= 2.0 ** 20
    w2[1024:] *= 2.0 ** -20
    out["weights_jump"] = stats(x, w2, m, 256)
    w3 = w.copy()
    w3[256:512] = 0.0
    out["zero_chunk"] = stats(x, w3, m, 256)
    x4 = x.copy()
    x4[700, np.isfinite(x4[700])] *= 1.0e6
    out["outlier_chunk"] = stats(x4, w, m, 256)
    for name, arr in (("x", x), ("w", w), ("w2", w2), ("w3", w3), ("x4_row", x4[700])):
        out["in_" + name] = arr
    out["in_sigma"], out["in_c"], out["in_mean"] = np.array(m[0]), m[1], m[2]
    x, m = data(rng, 900, 300, 64)
    out["k64"] = stats(x, None, m, 192)
    out["k64_x"], out["k64_sigma"], out["k64_c"], out["k64_mean"] = x, np.array(m[0]), m[1], m[2]
    x, m = data(rng, 1300, 1100, 20)
    out["d1100"] = stats(x, None, m, 320)
    out["d1100_x"], out["d1100_sigma"], out["d1100_c"], out["d1100_mean"] = x, np.array(m[0]), m[1], m[2]
    os.environ.pop("PPCA_GEN_CHUNK", None)
    np.savez(sys.argv[1], **out)
    print("wpred check written", sys.argv[1])


if __name__ == "__main__":
    main()
