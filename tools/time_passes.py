"""Times the output passes that share the per-sample core with the EM pass (llk, llks, infer, smooth,
extrapolate, covariance diagonal) through the public API, device-resident inputs (diagnostic)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ppca_rs_amd as P
from ppca_rs_amd import _lib

n, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.default_context()
truth = P.PPCAModel(0.1, np.random.default_rng(1).standard_normal((d, k)), np.random.default_rng(2).standard_normal(d))
spec = _lib.SynthSpec(0, n, d, k, 0.1, 0.3, 0, 0, 1033, truth._c.ctypes.data_as(_lib.c_double_p),
                      truth._mean.ctypes.data_as(_lib.c_double_p))
h = C.c_void_p()
_lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
ds = P.Dataset._wrap(h, ctx)
m = P.PPCAModel.init(k, ds, seed=3).iterate(ds).iterate(ds)
L = _lib.lib()
md = m._device(ctx)

def timed(name, fn, bytes_per_sample, reps=5):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:28s} {dt*1e3:9.2f} ms   {n/dt/1e6:8.1f} Msamples/s   {n*bytes_per_sample/dt/1e9:8.1f} GB/s algorithmic", flush=True)
    return out

tot = C.c_double()
timed("llk (total)", lambda: L.ppca_llk(ctx.handle, ds._h, md.h, C.byref(tot), None), 8 * d)
def recon(mode):
    o = C.c_void_p()
    _lib.check(L.ppca_reconstruct(ctx.handle, ds._h, md.h, mode, C.byref(o)))
    L.ppca_dataset_free(o)
timed("smooth (N x d out)", lambda: recon(0), 16 * d)
timed("extrapolate (N x d out)", lambda: recon(1), 16 * d)
def cdiag(mode):
    o = C.c_void_p()
    _lib.check(L.ppca_covariance_diagonal(ctx.handle, ds._h, md.h, mode, C.byref(o)))
    L.ppca_dataset_free(o)
timed("smoothed cov diagonal", lambda: cdiag(0), 16 * d)
nn = min(n, 1_000_000)
sub = ds._slice(0, nn)
st, cv = np.empty((nn, k)), np.empty((nn, k, k))
t0 = time.perf_counter(); _lib.check(L.ppca_infer(ctx.handle, sub._h, md.h, _lib.ptr(st), None)); t1 = time.perf_counter()
_lib.check(L.ppca_infer(ctx.handle, sub._h, md.h, _lib.ptr(st), _lib.ptr(cv))); t2 = time.perf_counter()
print(f"infer states only ({nn} rows, incl. D2H)      {1e3*(t1-t0):8.2f} ms")
print(f"infer states + covariances (incl. D2H {cv.nbytes/1e6:.0f} MB) {1e3*(t2-t1):8.2f} ms")
