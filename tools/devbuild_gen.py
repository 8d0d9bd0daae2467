"""Kernel-tuning build for ppca_generic.hip (the split pipeline: config 4, every shape outside the fused kernels): the file compiled
with the given -D flags and linked with the already-built objects of the other sources into
ppca_rs_amd/libppca_hip_<name>.so (select it with PPCA_HIP_LIB).

    python tools/devbuild_gen.py --name=pipe0 -DI8_PIPE=0
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "ppca_rs_amd", "csrc")
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
name = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--name=")]
name = name[0] if name else "gen"
obj = "/tmp/ppca_generic.%s.o" % name
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "-c", os.path.join(C, "ppca_generic.hip"), "-o", obj])
out = os.path.join(ROOT, "ppca_rs_amd", "libppca_hip_%s.so" % name)
objs = [obj] + [os.path.join(C, f) for f in ("ppca_kernels.o", "ppca_em8.o", "ppca_em9.o", "ppca_em16.o", "ppca_llk.o", "ppca_solve4.o", "ppca_comm.o", "ppca_capi.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
