set -x
mkdir -p gpurun_out/r2a
NCCL_DEBUG=WARN python -m pytest tests -m gpu -x -q -k "rccl or two_ranks or concurrent" > gpurun_out/r2a/test_rccl.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/test_rccl.log
tail -30 gpurun_out/r2a/test_rccl.log
# plain C++ host: system runtime + system rccl
python - <<'PY'
import ctypes as C, os
os.environ["PPCA_SYSTEM_HIP"]="1"
import sys; sys.path.insert(0,'.')
from ppca_rs_amd import _lib
l=_lib.lib()
print("backend:", l.ppca_comm_backend())
ctx=_lib.default_context()
buf=(C.c_char*128)()
_lib.check(l.ppca_comm_unique_id(buf))
h=C.c_void_p()
_lib.check(l.ppca_comm_create(ctx.handle,1,0,buf,C.byref(h)))
print("system-runtime comm ok, ranks", l.ppca_comm_n_ranks(h))
l.ppca_comm_destroy(h)
PY
