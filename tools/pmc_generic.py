"""PMC helper for the generic pipeline (d = 1024, k = 64, block masks as BASELINE config 4): a few EM steps on PMC_N
samples.  Run under `rocprofv3 --pmc ... --kernel-trace` (counter passes on their own, no trace domains besides
--kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P

n, d, k = int(os.environ.get("PMC_N", 200_000)), 1024, 64
truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
ds = truth.sample(n, 0.5, seed=3)
m = P.PPCAModel.init(k, ds, seed=4)
m = m.iterate(ds)
print("marker", ds.empty_dimensions())  # (tools/make_traffic_all.py sums the kernels between the two markers; they are its calibration)
for _ in range(int(os.environ.get("PMC_STEPS", 2))):
    m = m.iterate(ds)
print("marker", ds.empty_dimensions())
print("sigma", m.isotropic_noise)
