"""PMC helper: one dataset, one calibration kernel with a known byte count in the same access width
as the EM pass (column_presence_kernel: every element of X read once, 8 B/lane coalesced), then EM steps.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) with --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P

n, d, k = int(os.environ.get("PMC_N", 2_000_000)), 256, 10
truth = P.PPCAModel(0.1, np.random.default_rng(1011).standard_normal((d, k)), np.random.default_rng(1012).standard_normal(d))
ds = truth.sample(n, 0.3, seed=1013)
print("empty dims", ds.empty_dimensions(), "known X bytes", n * d * 8)
m = P.PPCAModel.init(k, ds, seed=2011)
for _ in range(3):
    m = m.iterate(ds)
print("sigma", m.isotropic_noise)
