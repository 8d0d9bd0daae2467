"""Kernel-tuning build: ppca_kernels.hip and ppca_llk.hip with -DPPCA_DEV_K10 (only the k = 10 int8-Gram variants,
~10x faster to compile) linked with the already-built objects of the other sources into
ppca_rs_amd/libppca_hip_dev.so (select it with PPCA_HIP_LIB).  `--timing` adds -DPPCA_PHASE_TIMING; `--asm` also
leaves /tmp/k10.s and /tmp/llk10.s; further -D flags are passed through."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "ppca_rs_amd", "csrc")
flags = ["-DPPCA_DEV_K10"] + [a for a in sys.argv[1:] if a.startswith("-D")] + (["-DPPCA_PHASE_TIMING"] if "--timing" in sys.argv else [])
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags]
procs = [subprocess.Popen(base + ["-c", os.path.join(C, "ppca_kernels.hip"), "-o", "/tmp/ppca_kernels.dev.o"]),
         subprocess.Popen(base + ["-c", os.path.join(C, "ppca_em8.hip"), "-o", "/tmp/ppca_em8.dev.o"]),
         subprocess.Popen(base + ["-c", os.path.join(C, "ppca_em9.hip"), "-o", "/tmp/ppca_em9.dev.o"]),
         subprocess.Popen(base + ["-c", os.path.join(C, "ppca_llk.hip"), "-o", "/tmp/ppca_llk.dev.o"])]
if "--timing" in sys.argv:  # ppca_capi prints the phase table only when built with the flag
    procs.append(subprocess.Popen(base + ["-c", os.path.join(C, "ppca_capi.hip"), "-o", "/tmp/ppca_capi.dev.o"]))
if "--asm" in sys.argv:
    for src, out in (("ppca_kernels.hip", "/tmp/k10.s"), ("ppca_em8.hip", "/tmp/em8.s"), ("ppca_llk.hip", "/tmp/llk10.s")):
        procs.append(subprocess.Popen(base + ["--offload-device-only", "-S", os.path.join(C, src), "-o", out], stderr=subprocess.DEVNULL))
assert all(p.wait() == 0 for p in procs)
capi = "/tmp/ppca_capi.dev.o" if "--timing" in sys.argv else os.path.join(C, "ppca_capi.o")
name = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--name=")]
out = os.path.join(ROOT, "ppca_rs_amd", "libppca_hip_%s.so" % (name[0] if name else "dev"))
subprocess.check_call(base[:2] + ["-shared", "-fPIC", "-o", out, "/tmp/ppca_kernels.dev.o", "/tmp/ppca_em8.dev.o", "/tmp/ppca_em9.dev.o", "/tmp/ppca_llk.dev.o",
                                  os.path.join(C, "ppca_generic.o"), os.path.join(C, "ppca_solve4.o"), os.path.join(C, "ppca_em16.o"), os.path.join(C, "ppca_comm.o"), capi])
print(out)
