"""PMC helper for the mixture (BASELINE config 5's shape at PMC_N rows: 8 components, d = 256, k = 10, 30 % masked): one
calibration kernel of known traffic (column_presence_kernel via empty_dimensions), 10 warm-up iterations from PPCAMix.init
(so that the responsibilities have formed, as in bench.py --config 5), then PMC_STEPS iterations inside a marker pair of
calibration launches: the counters of the kernels BETWEEN the second and third column_presence_kernel are the steady state.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) with --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P

n, d, k, nm = int(os.environ.get("PMC_N", 1_000_000)), 256, 10, 8
steps = int(os.environ.get("PMC_STEPS", 2))
blk = 65536
parts = []
for b0 in range(0, n, blk):
    c = (b0 // blk) % nm
    truth = P.PPCAModel(0.1, np.random.default_rng(1051 + 10 * c).standard_normal((d, k)), 3.0 * np.random.default_rng(1052 + 10 * c).standard_normal(d))
    parts.append(truth.sample(min(blk, n - b0), 0.3, seed=1053 + b0))
ds = P.Dataset.concat(parts)
del parts
mix = P.PPCAMix([P.PPCAModel(1.0, np.random.default_rng(2051 + c).standard_normal(d * k).reshape((k, d)).T.copy(), np.zeros(d)) for c in range(nm)], np.zeros(nm))
for _ in range(int(os.environ.get("PMC_WARM", 10))):
    mix = mix.iterate(ds)
print("marker", ds.empty_dimensions())
for _ in range(steps):
    mix = mix.iterate(ds)
print("marker", ds.empty_dimensions())
print("llk", mix.llk(ds))
