"""Times EM iterations of an arbitrary (N, d, k) configuration through the public API (diagnostic)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib
from ppca_rs_amd.distributed import ShardedEM

n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mask_kind, mask_run, mask_prob = (1, d // 2, 0.0) if len(sys.argv) > 4 and sys.argv[4] == "block" else (0, 0, 0.3)
import torch
torch.cuda.set_device(0)
ctx = _lib.default_context()
truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
spec = _lib.SynthSpec(0, n, d, k, 0.1, mask_prob, mask_kind, mask_run, 1033,
                      truth._c.ctypes.data_as(_lib.c_double_p), truth._mean.ctypes.data_as(_lib.c_double_p))
h = C.c_void_p()
t0 = time.time()
_lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
ds = P.Dataset._wrap(h, ctx)
print(f"generated {n}x{d} in {time.time()-t0:.2f}s; path kind {_lib.lib().ppca_path_kind(d, k)}", flush=True)
em = ShardedEM(ds, P.PPCAModel.init(k, ds, seed=3))
llks = []
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    em.step()
    torch.cuda.synchronize(); dt = time.time() - t0
    llks.append(em.llk_of_previous() / n)
    print(f"iter {it}: {dt*1e3:.1f} ms, llk/N of input model {llks[-1]:.4f}", flush=True)
assert all(b >= a for a, b in zip(llks, llks[1:])), llks
