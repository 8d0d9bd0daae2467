"""Discover the lane maps of v_mfma_i32_16x16x64_i8 with one-hot operands (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ppca_rs_amd import _lib
ctx = _lib.default_context()
def run(a, b):
    out = np.zeros((64, 4), dtype=np.int32)
    _lib.check(_lib.lib().ppca_debug_mfma_i8_probe(ctx.handle, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out)))
    return out
# hypothesis: A[row = l&15][k = 16*(l>>4) + j], B[k = 16*(l>>4)+j][col = l&15], D[row = 4*(l>>4)+r][col = l&15]
rng = np.random.default_rng(0)
A = rng.integers(-5, 6, (16, 64)).astype(np.int8)
B = rng.integers(-5, 6, (64, 16)).astype(np.int8)
a = np.zeros((64, 16), np.int8); b = np.zeros((64, 16), np.int8)
for l in range(64):
    for j in range(16):
        a[l, j] = A[l & 15, 16 * (l >> 4) + j]
        b[l, j] = B[16 * (l >> 4) + j, l & 15]
out = run(a, b)
D = np.zeros((16, 16), np.int64)
for l in range(64):
    for r in range(4):
        D[4 * (l >> 4) + r, l & 15] = out[l, r]
want = A.astype(np.int64) @ B.astype(np.int64)
print("hypothesis holds:", np.array_equal(D, want))
if not np.array_equal(D, want):
    # one-hot probing: which (lane, byte) of A pairs with which (lane, byte) of B, and where results land
    for la, ja in [(0, 0), (0, 1), (0, 8), (16, 0), (1, 0), (33, 5)]:
        a = np.zeros((64, 16), np.int8); a[la, ja] = 1
        b = np.ones((64, 16), np.int8)
        o = run(a, b)
        nz = np.argwhere(o != 0)
        print(f"A one-hot lane {la} byte {ja}: nonzero out (lane,reg) count {len(nz)} first {nz[:6].tolist()} vals {o[o!=0][:4]}")
    for lb, jb in [(0, 0), (0, 8), (16, 0), (1, 0)]:
        b = np.zeros((64, 16), np.int8); b[lb, jb] = 1
        a = np.ones((64, 16), np.int8)
        o = run(a, b)
        nz = np.argwhere(o != 0)
        print(f"B one-hot lane {lb} byte {jb}: nonzero out count {len(nz)} first {nz[:6].tolist()}")
    # pairing: A one-hot (lane 0, byte ja) x B one-hot (lane lb, byte jb) nonzero?
    for ja in (0, 1, 8, 15):
        hits = []
        for lb in (0, 16, 32, 48):
            for jb in range(16):
                a = np.zeros((64, 16), np.int8); a[0, ja] = 1
                b = np.zeros((64, 16), np.int8); b[lb, jb] = 1
                if run(a, b).any(): hits.append((lb, jb))
        print(f"A(lane 0, byte {ja}) pairs with B {hits}")
