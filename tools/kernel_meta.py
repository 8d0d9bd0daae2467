"""Code-object metadata of the kernels in an object / shared library built by hipcc: registers, spills, scratch, LDS.

    python tools/kernel_meta.py ppca_rs_amd/csrc/ppca_em8.o [name-filter]
"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def meta(path):
    out = []
    with tempfile.TemporaryDirectory() as td:
        import glob, shutil
        cp = os.path.join(td, "in.o")
        shutil.copy(path, cp)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], stdout=subprocess.DEVNULL, cwd=td)
        co = glob.glob(os.path.join(td, "in.o.*gfx950"))[0]
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
        out.append(dict(name=name, agpr=blk.split()[0], vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"),
                        sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in meta(sys.argv[1]):
        if flt in m["name"]:
            print("%-70s vgpr %s agpr %s sgpr %s | spilled v %s s %s | scratch %s B | static LDS %s" % (
                m["name"][:70], m["vgpr"], m["agpr"], m["sgpr"], m["vspill"], m["sspill"], m["scratch"], m["lds"]))
