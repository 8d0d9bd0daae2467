"""Kernel-tuning build for ppca_em16.hip: one instantiation (-DPPCA_E16_ONLY=K, default 16) linked with the already
built objects of the other sources into ppca_rs_amd/libppca_hip_dev16.so (select it with PPCA_HIP_LIB).  `--timing`
adds -DPPCA_PHASE_TIMING (also to ppca_generic.hip, which prints the phase table); `--asm` leaves /tmp/em16.s;
further -D flags are passed through."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "ppca_rs_amd", "csrc")
k = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--k=")]
timing = "--timing" in sys.argv
flags = ["-DPPCA_E16_ONLY=%s" % (k[0] if k else "16")] + [a for a in sys.argv[1:] if a.startswith("-D")] + (["-DPPCA_PHASE_TIMING"] if timing else [])
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags]
procs = [subprocess.Popen(base + ["-c", os.path.join(C, "ppca_em16.hip"), "-o", "/tmp/ppca_em16.dev.o"])]
gen_src, gen_obj = os.path.join(C, "ppca_generic.hip"), "/tmp/ppca_generic.dev.o"
if timing and not (os.path.exists(gen_obj) and os.path.getmtime(gen_obj) > os.path.getmtime(gen_src)):
    procs.append(subprocess.Popen(base + ["-c", os.path.join(C, "ppca_generic.hip"), "-o", "/tmp/ppca_generic.dev.o"]))
if "--asm" in sys.argv:
    procs.append(subprocess.Popen(base + ["--offload-device-only", "-S", os.path.join(C, "ppca_em16.hip"), "-o", "/tmp/em16.s"], stderr=subprocess.DEVNULL))
assert all(p.wait() == 0 for p in procs)
name = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--name=")]
out = os.path.join(ROOT, "ppca_rs_amd", "libppca_hip_%s.so" % (name[0] if name else "dev16"))
objs = ["/tmp/ppca_em16.dev.o", "/tmp/ppca_generic.dev.o" if timing else os.path.join(C, "ppca_generic.o")]
objs += [os.path.join(C, f) for f in ("ppca_kernels.o", "ppca_em8.o", "ppca_em9.o", "ppca_solve4.o", "ppca_llk.o", "ppca_comm.o", "ppca_capi.o")]
subprocess.check_call(base[:2] + ["-shared", "-fPIC", "-o", out] + objs)
print(out)
