"""Instruction mix of one kernel of an ISA listing (hipcc -S), split at '; E8_MARK n' / '; PPCA_MARK n' comments.

    python tools/isa_mix.py /tmp/em8.s 'em8_kernelILi10ELb0ELb0'
Classes: valu (v_* except MFMA), mfma64 (v_mfma_f64*), mfma8 (v_mfma_i32*), lds (ds_*), vmem (buffer_/global_/scratch_),
salu (s_* except waitcnt/nop/barrier/branch), wait (s_waitcnt/s_nop/s_barrier/s_sleep), branch."""
import re, sys, collections


def cls(op):
    if op.startswith("v_mfma_f64"): return "mfma64"
    if op.startswith("v_mfma"): return "mfma8"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main(path, sym):
    inside = False
    seg = "entry"
    order = []
    counts = collections.OrderedDict()
    for line in open(path):
        t = line.strip()
        if not inside:
            m0 = re.match(r"^([A-Za-z_]\S*):", t)
            if m0 and sym in m0.group(1):
                inside = True
            continue
        if t.startswith(".end_amdhsa_kernel") or t.startswith("s_endpgm") and False:
            break
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r";\s*(E8_MARK|PPCA_MARK|E16_MARK)\s+(\S+)", t)
        if m:
            seg = "after mark " + m.group(2) + " #%d" % len(counts)
            continue
        if t.startswith(".LBB") and t.endswith(":"):
            seg = seg.split(" @")[0] + " @" + t[:-1]
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = counts.setdefault(seg, collections.Counter())
        c[cls(op)] += 1
    tot = collections.Counter()
    for seg, c in counts.items():
        n = sum(c.values())
        if n >= 8:
            print("%-40s %5d  " % (seg[-40:], n) + "  ".join("%s %d" % kv for kv in sorted(c.items())))
        tot.update(c)
    print("TOTAL", dict(tot))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
