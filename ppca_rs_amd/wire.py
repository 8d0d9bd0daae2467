"""bincode layouts of the reference's serde structs, so that `dump()` bytes can travel between the
reference package and this one.

PARITY UNPINNED: the reference cannot be built here and holds no stored artefact, so these layouts are written from
the published serde implementations of the pinned crates and have NOT been checked against bytes produced by the real
`ppca_rs`:
  * bincode 1.3.3 default options (src/python_bindings.rs:66-79, :388-401 call `bincode::serialize/deserialize`):
    little-endian, fixed-width integers, `usize`/lengths as u64, structs = their fields in order, `Arc<T>`/`Cow<T>` = T
    (serde "rc" feature, ppca/Cargo.toml:31), unit = nothing.
  * nalgebra 0.32.2 `serde-serialize`: a matrix serialises its storage, `VecStorage {data: Vec<T> (column-major),
    nrows, ncols}`; `Dyn(n)` = n as usize, `Const<N>` = unit.  So DMatrix = [len][data..][nrows][ncols] and
    DVector = [len][data..][nrows].
  * bit-vec 0.6.3 `serde`: `BitVec<u32> {storage: Vec<u32>, nbits: usize}`, bit i = bit (i % 32) of block i / 32.
Structs: `PPCAModelInner {output_covariance: OutputCovariance {isotropic_noise: f64, transform: DMatrix}, mean:
DVector}` (ppca_model.rs:18-22, output_covariance.rs:18-24); `Dataset {data: Arc<Vec<MaskedSample {data: DVector, mask:
Mask(BitVec)}>>, weights: Vec<f64>}` (dataset.rs:10-14, :92-100); `PPCAMixInner {output_size: usize, models:
Vec<PPCAModel>, log_weights: DVector}` (mix.rs:27-37).
"""
from __future__ import annotations

import struct
from typing import List, Tuple

import numpy as np

U64 = struct.Struct("<Q")
F64 = struct.Struct("<d")


def _vec_f64(a: np.ndarray) -> bytes:
    a = np.ascontiguousarray(a, dtype="<f8").ravel()
    return U64.pack(a.size) + a.tobytes()


def _dmatrix(m: np.ndarray) -> bytes:
    m = np.asarray(m, dtype=np.float64)
    return _vec_f64(m.T) + U64.pack(m.shape[0]) + U64.pack(m.shape[1])  # column-major data, nrows, ncols


def _dvector(v: np.ndarray) -> bytes:
    v = np.asarray(v, dtype=np.float64).ravel()
    return _vec_f64(v) + U64.pack(v.size)


class _Reader:
    def __init__(self, data: bytes):
        self.b, self.o = memoryview(data), 0

    def _need(self, nbytes: int) -> None:
        if self.o + nbytes > len(self.b):
            raise ValueError("io error: unexpected end of file")  # bincode's message for a short buffer

    def u64(self) -> int:
        self._need(8)
        (v,) = U64.unpack_from(self.b, self.o)
        self.o += 8
        return v

    def f64(self) -> float:
        self._need(8)
        (v,) = F64.unpack_from(self.b, self.o)
        self.o += 8
        return v

    def vec_f64(self) -> np.ndarray:
        n = self.u64()
        self._need(8 * n)
        a = np.frombuffer(self.b, dtype="<f8", count=n, offset=self.o).copy()
        self.o += 8 * n
        return a

    def dmatrix(self) -> np.ndarray:
        data = self.vec_f64()
        r, c = self.u64(), self.u64()
        if r * c != data.size:
            raise ValueError("invalid matrix: data length does not match its shape")
        return data.reshape(c, r).T.copy()

    def dvector(self) -> np.ndarray:
        data = self.vec_f64()
        if self.u64() != data.size:
            raise ValueError("invalid vector: data length does not match its shape")
        return data

    def done(self) -> None:
        if self.o != len(self.b):
            raise ValueError("trailing bytes after the value")


def dump_model(sigma: float, transform: np.ndarray, mean: np.ndarray) -> bytes:
    return F64.pack(float(sigma)) + _dmatrix(transform) + _dvector(mean)


def _read_model(r: _Reader) -> Tuple[float, np.ndarray, np.ndarray]:
    sigma = r.f64()
    c = r.dmatrix()
    return sigma, c, r.dvector()


def load_model(data: bytes) -> Tuple[float, np.ndarray, np.ndarray]:
    r = _Reader(data)
    out = _read_model(r)
    r.done()
    return out


def dump_mix(models: List[Tuple[float, np.ndarray, np.ndarray]], log_weights: np.ndarray) -> bytes:
    out = U64.pack(models[0][1].shape[0]) + U64.pack(len(models))
    for m in models:
        out += dump_model(*m)
    return out + _dvector(log_weights)


def load_mix(data: bytes):
    r = _Reader(data)
    r.u64()  # output_size
    models = [_read_model(r) for _ in range(r.u64())]
    lw = r.dvector()
    r.done()
    return models, lw


def dump_dataset(x: np.ndarray, weights: np.ndarray) -> bytes:
    """x: (n, d) with non-finite = masked (the stored vector keeps the value, the mask bit is cleared,
    dataset.rs:19-22)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, d = x.shape
    nblk = (d + 31) // 32
    rec = np.dtype([("len", "<u8"), ("data", "<f8", (d,)), ("nrows", "<u8"), ("nblk", "<u8"), ("blk", "<u4", (nblk,)),
                    ("nbits", "<u8")])
    out = np.zeros(n, dtype=rec)
    out["len"], out["nrows"], out["nblk"], out["nbits"] = d, d, nblk, d
    out["data"] = x.reshape(n, d)
    bits = np.zeros((n, nblk * 32), dtype=np.uint8)
    bits[:, :d] = np.isfinite(x)
    out["blk"] = np.packbits(bits.reshape(n, nblk, 32), axis=2, bitorder="little").view("<u4").reshape(n, nblk)
    return U64.pack(n) + out.tobytes() + _vec_f64(weights)


def load_dataset(data: bytes) -> Tuple[np.ndarray, np.ndarray]:
    r = _Reader(data)
    n = r.u64()
    if n == 0:
        w = r.vec_f64()
        r.done()
        return np.zeros((0, 0)), w
    d = U64.unpack_from(r.b, r.o)[0]
    nblk = (d + 31) // 32
    rec = np.dtype([("len", "<u8"), ("data", "<f8", (d,)), ("nrows", "<u8"), ("nblk", "<u8"), ("blk", "<u4", (nblk,)),
                    ("nbits", "<u8")])
    if r.o + n * rec.itemsize > len(r.b):
        raise ValueError("io error: unexpected end of file")
    arr = np.frombuffer(r.b, dtype=rec, count=n, offset=r.o)
    if not (np.all(arr["len"] == d) and np.all(arr["nrows"] == d) and np.all(arr["nblk"] == nblk) and np.all(arr["nbits"] == d)):
        raise ValueError("samples of different sizes are not supported")
    r.o += n * rec.itemsize
    bits = np.unpackbits(np.ascontiguousarray(arr["blk"]).view(np.uint8).reshape(n, nblk * 4), axis=1, bitorder="little")[:, :d]
    x = np.where(bits.astype(bool), arr["data"].reshape(n, d), np.nan)
    w = r.vec_f64()
    r.done()
    return x, w
