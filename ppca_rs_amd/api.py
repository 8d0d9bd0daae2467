"""Python mirror of the reference's class surface for the EM hot path.

Same names, argument meaning and error behaviour as the PyO3 module
`ppca_rs.ppca_rs` (reference: src/python_bindings.rs:15-26) and the trainers of
python/ppca_rs/__init__.py, built over the C-ABI of include/ppca_hip.h.  All heavy
calls run on the GPU through libppca_hip.so; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import io
import math
from dataclasses import dataclass
from typing import Iterator, List, Literal, Optional, Sequence

import numpy as np

from . import _lib
from . import wire as _wire
from ._lib import check, f64, lib, ptr


def _ctx(ctx=None) -> _lib.Context:
    return ctx if ctx is not None else _lib.default_context()


# --------------------------------------------------------------------------- Dataset
class Dataset:
    """A device-resident dataset (reference: `Dataset`, src/python_bindings.rs:28-134).

    `Dataset(ndarray, weights=None)`: float64 (N, d); non-finite entries are masked
    (dataset.rs:19-22); weights default to 1 (dataset.rs:153-158).
    """

    def __init__(self, ndarray, weights=None, *, ctx=None, _handle=None):
        self._ctx = _ctx(ctx)
        if _handle is not None:
            self._h = _handle
            return
        arr = np.asarray(ndarray)
        if arr.dtype != np.float64:
            raise TypeError("Dataset expects a float64 array (reference: PyReadonlyArray2<f64>)")
        if arr.ndim != 2:
            raise TypeError("Dataset expects a 2-D array (n_samples, n_features)")
        w = None
        if weights is not None:
            w = np.ascontiguousarray(np.asarray(weights), dtype=np.float64).ravel()
            if w.shape[0] != arr.shape[0]:
                raise ValueError("weights and data differ in length")  # assert_eq! dataset.rs:162
        if arr.shape[1] < 1:
            raise ValueError("Dataset needs at least one feature")
        h = C.c_void_p()
        es = arr.itemsize
        check(lib().ppca_dataset_from_host(self._ctx.handle, C.c_void_p(arr.ctypes.data), arr.shape[0], arr.shape[1],
                                           arr.strides[0] // es, arr.strides[1] // es, ptr(w), C.byref(h)))
        self._h = h

    @classmethod
    def _wrap(cls, handle, ctx) -> "Dataset":
        return cls(None, ctx=ctx, _handle=handle)

    @classmethod
    def from_device(cls, x_ptr: int, n: int, d: int, weights_ptr: int | None = None, *, ctx=None, keepalive=None):
        """Borrow device memory (e.g. a torch tensor's data_ptr()); `keepalive` is held."""
        c = _ctx(ctx)
        h = C.c_void_p()
        check(lib().ppca_dataset_from_device(c.handle, C.c_void_p(x_ptr), n, d,
                                             C.c_void_p(weights_ptr) if weights_ptr else None, C.byref(h)))
        ds = cls._wrap(h, c)
        ds._keepalive = keepalive
        return ds

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ppca_dataset_free(self._h)
                self._h = None
        except Exception:
            pass

    def __len__(self) -> int:
        return int(lib().ppca_dataset_len(self._h))

    def output_size(self) -> Optional[int]:
        """dataset.rs:189-191 -- None for an empty dataset."""
        return int(lib().ppca_dataset_output_size(self._h)) if len(self) > 0 else None

    @property
    def _d(self) -> int:
        return int(lib().ppca_dataset_output_size(self._h))

    def numpy(self) -> np.ndarray:
        """(N, d) float64 with NaN at masked positions (src/python_bindings.rs:81-92)."""
        out = np.empty((len(self), self._d), dtype=np.float64)
        check(lib().ppca_dataset_to_host(self._h, ptr(out)))
        return out

    def weights(self) -> np.ndarray:
        out = np.empty(len(self), dtype=np.float64)
        check(lib().ppca_dataset_weights_to_host(self._h, ptr(out)))
        return out

    def with_weights(self, weights) -> "Dataset":
        """Same rows, new weights, no copy of the rows (dataset.rs:171-176)."""
        w = np.ascontiguousarray(np.asarray(weights), dtype=np.float64).ravel()
        if w.shape[0] != len(self):
            raise ValueError("weights and data differ in length")
        h = C.c_void_p()
        check(lib().ppca_dataset_with_weights(self._h, ptr(w), None, C.byref(h)))
        return Dataset._wrap(h, self._ctx)

    def empty_dimensions(self) -> List[int]:
        """Dimensions masked in every sample (dataset.rs:194-222)."""
        if len(self) == 0:
            return []
        flags = (C.c_int32 * self._d)()
        check(lib().ppca_dataset_empty_dimensions(self._h, flags))
        return [j for j in range(self._d) if flags[j]]

    def chunks(self, chunks: int) -> "DatasetChunks":
        """Iterator over ceil(N / chunks)-row slices (src/python_bindings.rs:110-118)."""
        return DatasetChunks(self, chunks)

    def _slice(self, start: int, length: int) -> "Dataset":
        h = C.c_void_p()
        check(lib().ppca_dataset_slice(self._h, start, length, C.byref(h)))
        ds = Dataset._wrap(h, self._ctx)
        ds._parent = self
        return ds

    @staticmethod
    def concat(datasets: Sequence["Dataset"]) -> "Dataset":
        """src/python_bindings.rs:121-133"""
        datasets = list(datasets)
        if not datasets:
            raise ValueError("cannot concatenate an empty list of datasets")
        ctx = datasets[0]._ctx
        arr = (C.c_void_p * len(datasets))(*[d._h for d in datasets])
        h = C.c_void_p()
        check(lib().ppca_dataset_concat(ctx.handle, arr, len(datasets), C.byref(h)))
        return Dataset._wrap(h, ctx)

    # dump/load (src/python_bindings.rs:66-79).  Default container: npz (round trip verified here);
    # format="bincode" writes the reference's bincode layout as restated in wire.py (parity unpinned: no real
    # artefact to check against).  load() accepts both.
    def dump(self, format: str = "npz") -> bytes:
        if format == "bincode":
            return _wire.dump_dataset(self.numpy(), self.weights())
        buf = io.BytesIO()
        np.savez(buf, kind="ppca_rs_amd.Dataset", data=self.numpy(), weights=self.weights())
        return buf.getvalue()

    @staticmethod
    def load(data: bytes) -> "Dataset":
        try:
            if bytes(data[:2]) != b"PK":
                x, w = _wire.load_dataset(data)
                return Dataset(x, w)
            z = np.load(io.BytesIO(data), allow_pickle=False)
            return Dataset(z["data"], z["weights"])
        except Exception as err:  # reference: bincode error -> Exception(str)
            raise Exception(str(err))

    def __getstate__(self):
        return self.dump()

    def __setstate__(self, state):
        other = Dataset.load(state)
        self._ctx, self._h = other._ctx, other._h
        other._h = None

    def __repr__(self):
        return f"Dataset(n={len(self)}, output_size={self.output_size()})"


class DatasetChunks:
    """src/python_bindings.rs:136-166"""

    def __init__(self, dataset: Dataset, chunks: int):
        if chunks <= 0:
            raise ValueError("chunks must be positive")
        self.dataset = dataset
        self.length = len(dataset)
        self.stride = int(math.ceil(self.length / chunks))
        self.position = 0

    def __iter__(self) -> Iterator[Dataset]:
        return self

    def __next__(self) -> Dataset:
        if self.position < self.length:
            n = min(self.length, self.position + self.stride) - self.position
            out = self.dataset._slice(self.position, n)
            self.position += self.stride
            return out
        raise StopIteration


# --------------------------------------------------------------------------- Prior
class Prior:
    """MAP priors (prior.rs:8-65; src/python_bindings.rs:168-201).  Builders return a new Prior."""

    def __init__(self):
        self.mean: Optional[np.ndarray] = None
        self.mean_covariance: Optional[np.ndarray] = None
        self.isotropic_noise_alpha: Optional[float] = None
        self.isotropic_noise_beta: Optional[float] = None
        self.transformation_precision: float = 0.0

    def _copy(self) -> "Prior":
        p = Prior()
        p.__dict__.update(self.__dict__)
        return p

    def with_mean_prior(self, mean, mean_covariance) -> "Prior":
        mean = f64(mean).ravel()  # the reference wants 2-D row/column; 1-D is accepted too
        cov = f64(mean_covariance)
        if cov.shape != (mean.shape[0], mean.shape[0]):
            raise ValueError("mean covariance must be (d, d)")  # assert_eq! prior.rs:33-34
        if not np.isfinite(np.linalg.cond(cov)) or np.linalg.matrix_rank(cov) < cov.shape[0]:
            raise ValueError("mean covariance should be invertible")  # prior.rs:40
        p = self._copy()
        p.mean, p.mean_covariance = mean, cov
        return p

    def with_isotropic_noise_prior(self, alpha: float, beta: float) -> "Prior":
        if not (alpha >= 0.0 and beta >= 0.0):
            raise ValueError("alpha and beta must be >= 0")  # prior.rs:50-51
        p = self._copy()
        p.isotropic_noise_alpha, p.isotropic_noise_beta = float(alpha), float(beta)
        return p

    def with_transformation_precision(self, precision: float) -> "Prior":
        if not precision >= 0.0:
            raise ValueError("precision must be >= 0")  # prior.rs:61
        p = self._copy()
        p.transformation_precision = float(precision)
        return p

    def _c(self):
        c = _lib.Prior()
        c.has_mean_prior = int(self.mean is not None)
        if self.mean is not None:
            c.mean = self.mean.ctypes.data_as(_lib.c_double_p)
            c.mean_covariance = self.mean_covariance.ctypes.data_as(_lib.c_double_p)
        c.has_isotropic_noise_prior = int(self.isotropic_noise_alpha is not None)
        c.isotropic_noise_alpha = self.isotropic_noise_alpha or 0.0
        c.isotropic_noise_beta = self.isotropic_noise_beta or 0.0
        c.transformation_precision = self.transformation_precision
        return c


def _prior_ref(prior: Optional[Prior]):
    if prior is None:
        return None, None
    c = prior._c()
    return C.byref(c), c


# --------------------------------------------------------------------------- PPCAModel
class _DevModel:
    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        try:
            if self.h:
                lib().ppca_model_free(self.h)
                self.h = None
        except Exception:
            pass


class PPCAModel:
    """Immutable PPCA model (ppca_model.rs:18-48; src/python_bindings.rs:367-533).

    y = C x + mean + noise, x ~ N(0, I), noise ~ N(0, isotropic_noise^2 I).
    """

    def __init__(self, isotropic_noise: float, transform, mean, *, ctx=None):
        t = np.array(transform, dtype=np.float64, order="C")
        if t.ndim != 2:
            raise TypeError("transform must be a 2-D float64 array (d, k)")
        m = np.asarray(mean, dtype=np.float64)
        if m.ndim == 2 and 1 in m.shape:
            m = m.reshape(-1)
        elif m.ndim == 2:
            # to_nalgebra_vector panics: "Expected column- or row- vector" (src/utils.rs:16-22)
            raise ValueError(f"Expected column- or row- vector; got {m.shape[0]}x{m.shape[1]} matrix")
        elif m.ndim != 1:
            raise TypeError("mean must be a vector")
        if m.shape[0] != t.shape[0]:
            raise ValueError("mean and transform disagree on the output size")
        self._sigma = float(isotropic_noise)
        self._c = t
        self._mean = np.array(m, dtype=np.float64)
        self._c.setflags(write=False)
        self._mean.setflags(write=False)
        self._ctx = ctx
        self._dev: Optional[_DevModel] = None

    # -- device handle ------------------------------------------------------
    def _device(self, ctx) -> _DevModel:
        if self._dev is None or self._dev_ctx is not ctx:
            h = C.c_void_p()
            check(lib().ppca_model_create(ctx.handle, self.output_size, self.state_size, self._sigma, ptr(self._c),
                                          ptr(self._mean), C.byref(h)))
            self._dev, self._dev_ctx = _DevModel(h), ctx
        return self._dev

    @classmethod
    def _from_device(cls, dev: _DevModel, ctx, d: int, k: int) -> "PPCAModel":
        sig = C.c_double(0.0)
        c = np.empty((d, k))
        m = np.empty(d)
        check(lib().ppca_model_download(dev.h, C.byref(sig), ptr(c), ptr(m)))
        out = cls(sig.value, c, m)
        out._dev, out._dev_ctx = dev, ctx
        return out

    # -- getters (src/python_bindings.rs:403-447) ------------------------------
    @property
    def output_size(self) -> int:
        return int(self._c.shape[0])

    @property
    def state_size(self) -> int:
        return int(self._c.shape[1])

    @property
    def n_parameters(self) -> int:
        """ppca_model.rs:107-109"""
        return 1 + self.state_size * self.output_size + self.output_size

    @property
    def singular_values(self) -> np.ndarray:
        """sqrt of each column's norm (sic, ppca_model.rs:113-121)."""
        return np.sqrt(np.linalg.norm(self._c, axis=0))

    @property
    def transform(self) -> np.ndarray:
        return self._c.copy()

    @property
    def isotropic_noise(self) -> float:
        return self._sigma

    @property
    def mean(self) -> np.ndarray:
        return self._mean.copy()

    def __repr__(self) -> str:  # shape of src/python_bindings.rs:454-464
        return (f"PPCAModel(isotropic_noise={self._sigma}, transform=array({self._c}, dtype=\"float64\"), "
                f"mean=narray({self._mean}, dtype=\"float64\"))")

    # -- construction ---------------------------------------------------------
    @staticmethod
    def init(state_size: int, dataset: Dataset, seed: Optional[int] = None) -> "PPCAModel":
        """Random untrained model (ppca_model.rs:51-70): C ~ N(0,1) with the rows of
        all-masked dimensions zeroed, sigma = 1, mean = 0.  `seed` is an extension
        (the reference's RNG cannot be seeded).

        LIMIT: state_size <= 64.  The reference takes any state size; here the per-sample k x k inversion is one
        wave's job (a 64 x 64 matrix in 33 KB of LDS, ppca_generic.hip::solve_mfma_kernel) and the M-step row solves
        keep one row per lane.  Larger state sizes raise here instead of at the first kernel launch."""
        if len(dataset) == 0:
            raise ValueError("dataset is empty")  # assert!(!dataset.is_empty()) :52
        if state_size > 64:
            raise ValueError(f"state_size {state_size} is not supported: the MI355X kernels cover state sizes up to 64")
        if state_size < 0:
            raise ValueError("state_size must be >= 0")
        # state_size = 0 (an isotropic Gaussian around the mean, ppca_model.rs:51-70 with an empty transform; to_canonical
        # :399-402) is accepted as the reference accepts it: the library carries it as ONE zero transform column, with
        # which every pass reproduces the k = 0 model exactly (include/ppca_hip.h, ppca_model_create)
        d = dataset.output_size()
        rng = np.random.default_rng(seed)
        # DMatrix::from_vec is column-major (utils.rs:16-25)
        c = rng.standard_normal(d * state_size).reshape((state_size, d)).T.copy()
        for j in dataset.empty_dimensions():
            c[j, :] = 0.0
        return PPCAModel(1.0, c, np.zeros(d))

    def sample(self, dataset_size: int, mask_prob: float, seed: Optional[int] = None, *, ctx=None) -> Dataset:
        """ppca_model.rs:186-191, generated on the GPU with a counter-based RNG."""
        if not (0.0 <= mask_prob <= 1.0):
            raise ValueError("invalid mask probability")  # :171
        c = _ctx(ctx or self._ctx)
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1)[0])
        spec = _lib.SynthSpec(0, dataset_size, self.output_size, self.state_size, self._sigma, float(mask_prob), 0, 0,
                              seed, self._c.ctypes.data_as(_lib.c_double_p), self._mean.ctypes.data_as(_lib.c_double_p))
        h = C.c_void_p()
        check(lib().ppca_dataset_generate(c.handle, C.byref(spec), C.byref(h)))
        return Dataset._wrap(h, c)

    # -- hot path ---------------------------------------------------------------
    def llk(self, dataset: Dataset) -> float:
        """Weighted log-likelihood (ppca_model.rs:142-149)."""
        tot = C.c_double(0.0)
        check(lib().ppca_llk(dataset._ctx.handle, dataset._h, self._device(dataset._ctx).h, C.byref(tot), None))
        return tot.value

    def llks(self, dataset: Dataset) -> np.ndarray:
        """Per-sample log-likelihood (ppca_model.rs:152-159)."""
        out = np.empty(len(dataset))
        check(lib().ppca_llk(dataset._ctx.handle, dataset._h, self._device(dataset._ctx).h, None, ptr(out)))
        return out

    def infer(self, dataset: Dataset) -> "InferredMasked":
        """ppca_model.rs:221-227"""
        n, k = len(dataset), self.state_size
        states = np.empty((n, k))
        covs = np.empty((n, k, k))
        check(lib().ppca_infer(dataset._ctx.handle, dataset._h, self._device(dataset._ctx).h, ptr(states), ptr(covs)))
        return InferredMasked(self, states, covs)

    def _recon(self, dataset: Dataset, mode: int, fn) -> Dataset:
        h = C.c_void_p()
        check(fn(dataset._ctx.handle, dataset._h, self._device(dataset._ctx).h, mode, C.byref(h)))
        return Dataset._wrap(h, dataset._ctx)

    def smooth(self, dataset: Dataset) -> Dataset:
        """C z + mean for every dimension (ppca_model.rs:237-244)."""
        return self._recon(dataset, 0, lib().ppca_reconstruct)

    filter_extrapolate = smooth  # README name (readme.md:62)

    def extrapolate(self, dataset: Dataset) -> Dataset:
        """Observed values kept, masked ones replaced by C z + mean (ppca_model.rs:254-261)."""
        return self._recon(dataset, 1, lib().ppca_reconstruct)

    def _iterate(self, dataset: Dataset, prior: Optional[Prior], want_llk: bool):
        ctx = dataset._ctx
        if len(dataset) == 0:
            raise ValueError("dataset is empty")
        out = C.c_void_p()
        check(lib().ppca_model_alloc(ctx.handle, self.output_size, self.state_size, C.byref(out)))
        dev_out = _DevModel(out)
        pref, keep = _prior_ref(prior)
        llk = C.c_double(0.0)
        check(lib().ppca_em_step(ctx.handle, dataset._h, self._device(ctx).h, pref, dev_out.h,
                                 C.byref(llk) if want_llk else None))
        new = PPCAModel._from_device(dev_out, ctx, self.output_size, self.state_size)
        return new, (llk.value if want_llk else None)

    def iterate(self, dataset: Dataset) -> "PPCAModel":
        """One EM iteration (ppca_model.rs:267-269)."""
        return self._iterate(dataset, None, False)[0]

    def iterate_with_prior(self, dataset: Dataset, prior: Prior) -> "PPCAModel":
        """One MAP-EM iteration (ppca_model.rs:277-393)."""
        return self._iterate(dataset, prior, False)[0]

    def iterate_with_llk(self, dataset: Dataset, prior: Optional[Prior] = None):
        """Extension: (next model, llk of THIS model) from the same single pass."""
        return self._iterate(dataset, prior, True)

    def to_canonical(self) -> "PPCAModel":
        """C = U S V^T -> U S, columns by descending singular value, sign = signum(column sum)
        (ppca_model.rs:398-425).  Host-side, O(d k^2)."""
        if self.state_size == 0:
            return self
        u, s, _ = np.linalg.svd(self._c, full_matrices=False)
        c = u * s
        sums = c.sum(axis=0)
        c = c * np.where(np.signbit(sums), -1.0, 1.0)
        return PPCAModel(self._sigma, c, self._mean)

    # -- serialisation (own container; bincode layout is a "next" row) ------------
    def dump(self, format: str = "npz") -> bytes:
        """src/python_bindings.rs:394-401; format="bincode": the reference's layout (wire.py, parity unpinned)."""
        if format == "bincode":
            return _wire.dump_model(self._sigma, self._c, self._mean)
        buf = io.BytesIO()
        np.savez(buf, kind="ppca_rs_amd.PPCAModel", isotropic_noise=self._sigma, transform=self._c, mean=self._mean)
        return buf.getvalue()

    @staticmethod
    def load(data: bytes) -> "PPCAModel":
        try:
            if bytes(data[:2]) != b"PK":
                return PPCAModel(*_wire.load_model(data))
            z = np.load(io.BytesIO(data), allow_pickle=False)
            return PPCAModel(float(z["isotropic_noise"]), z["transform"], z["mean"])
        except Exception as err:
            raise Exception(str(err))

    def __getstate__(self):
        return self.dump()

    def __setstate__(self, state):
        o = PPCAModel.load(state)
        self.__dict__.update(o.__dict__)

    def __getnewargs__(self):
        return (self._sigma, self.transform, self.mean)


# --------------------------------------------------------------------------- InferredMasked
class InferredMasked:
    """Batch of per-sample posteriors (src/python_bindings.rs:203-345; ppca_model.rs:430-593)."""

    def __init__(self, model: PPCAModel, states: np.ndarray, covs: np.ndarray):
        self._model, self._states, self._covs = model, states, covs

    def states(self) -> np.ndarray:
        if self._states.shape[0] == 0:
            return np.zeros((0, 0))
        return self._states.copy()

    def covariances(self) -> List[np.ndarray]:
        return [c.copy() for c in self._covs]

    def _as_dataset(self, arr: np.ndarray) -> Dataset:
        return Dataset(np.ascontiguousarray(arr))

    def smoothed(self, ppca: PPCAModel) -> Dataset:
        """C z + mean (ppca_model.rs:454-456)."""
        return self._as_dataset(self._states @ ppca._c.T + ppca._mean)

    def extrapolated(self, ppca: PPCAModel, dataset: Dataset) -> Dataset:
        """ppca_model.rs:460-463"""
        x = dataset.numpy()
        sm = self._states @ ppca._c.T + ppca._mean
        return self._as_dataset(np.where(np.isfinite(x), x, sm))

    def smoothed_covariances(self, ppca: PPCAModel) -> List[np.ndarray]:
        """sigma^2 I + C Sigma C^T per sample (ppca_model.rs:471-477) -- d x d each."""
        d = ppca.output_size
        eye = np.eye(d) * ppca._sigma ** 2
        return [eye + ppca._c @ cv @ ppca._c.T for cv in self._covs]

    def smoothed_covariances_diagonal(self, ppca: PPCAModel) -> Dataset:
        """ppca_model.rs:485-508"""
        diag = np.einsum("ja,nab,jb->nj", ppca._c, self._covs, ppca._c) + ppca._sigma ** 2
        return self._as_dataset(diag)

    def extrapolated_covariances(self, ppca: PPCAModel, dataset: Dataset) -> List[np.ndarray]:
        """ppca_model.rs:517-534"""
        x = dataset.numpy()
        d = ppca.output_size
        out = []
        for cv, row in zip(self._covs, x):
            neg = ~np.isfinite(row)
            full = np.zeros((d, d))
            if neg.any():
                cn = ppca._c[neg]
                full[np.ix_(neg, neg)] = np.eye(neg.sum()) * ppca._sigma ** 2 + cn @ cv @ cn.T
            out.append(full)
        return out

    def extrapolated_covariances_diagonal(self, ppca: PPCAModel, dataset: Dataset) -> Dataset:
        """ppca_model.rs:542-577"""
        x = dataset.numpy()
        diag = np.einsum("ja,nab,jb->nj", ppca._c, self._covs, ppca._c) + ppca._sigma ** 2
        return self._as_dataset(np.where(np.isfinite(x), 0.0, diag))

    def posterior_sampler(self) -> "PosteriorSampler":
        """ppca_model.rs:581-592"""
        return PosteriorSampler(self._model, self._states, np.linalg.cholesky(self._covs))


class PosteriorSampler:
    """ppca_model.rs:597-626; src/python_bindings.rs:347-365"""

    def __init__(self, model: PPCAModel, states: np.ndarray, chol: np.ndarray):
        self._model, self._states, self._chol = model, states, chol

    def sample(self, seed: Optional[int] = None) -> Dataset:
        rng = np.random.default_rng(seed)
        n, k = self._states.shape
        m = self._model
        std = rng.standard_normal((n, k))
        noise = m._sigma * rng.standard_normal((n, m.output_size))
        z = self._states + np.einsum("nab,nb->na", self._chol, std)
        return Dataset(np.ascontiguousarray(noise + m._mean + z @ m._c.T))


# --------------------------------------------------------------------------- trainers
@dataclass(frozen=True)
class TrainMetrics:
    """python/ppca_rs/__init__.py:14-18"""
    llk: float
    aic: float
    bic: float


def _metrics(llk: float, n_parameters: int, n: int) -> TrainMetrics:
    # formulas as written in python/ppca_rs/__init__.py:52-57
    return TrainMetrics(llk=llk / n, aic=2.0 * (n_parameters - llk) / n, bic=(llk - n_parameters * np.log(n)) / n)


@dataclass
class PPCATrainer:
    """EM driver (python/ppca_rs/__init__.py:21-67)."""

    dataset: Dataset

    def train(self, *, start: Optional[PPCAModel] = None, prior: Optional[Prior] = None, state_size: int,
              n_iters: int = 10, metric: Literal["aic", "bic", "llk"] = "aic", quiet: bool = False,
              seed: Optional[int] = None) -> PPCAModel:
        model = start or PPCAModel.init(state_size, self.dataset, seed=seed)
        n = len(self.dataset)
        for idx in range(n_iters):
            if not quiet:
                # the llk of the current model is a by-product of the EM pass: no second sweep
                new_model, llk = model.iterate_with_llk(self.dataset, prior)
                metrics = _metrics(llk, model.n_parameters, n)
                print(f"Masked PPCA iteration {idx + 1}: {metric}={getattr(metrics, metric)}")
                model = new_model
            else:
                model = model.iterate_with_prior(self.dataset, prior) if prior is not None else model.iterate(self.dataset)
        return model.to_canonical()


# --------------------------------------------------------------------------- mixture
def _log_softmax(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, dtype=np.float64)
    mx = v.max()
    return v - mx - np.log(np.exp(v - mx).sum())


class PPCAMix:
    """Mixture of PPCA models (mix.rs:27-83; src/python_bindings.rs:535-711)."""

    def __init__(self, models: Sequence[PPCAModel], log_weights):
        models = list(models)
        lw = f64(log_weights).ravel()
        if len(models) == 0:
            raise ValueError("need at least one model")  # assert! mix.rs:51
        if len(models) != lw.shape[0]:
            raise ValueError("models and log_weights differ in length")  # mix.rs:52
        sizes = {m.output_size for m in models}
        if len(sizes) != 1:
            raise ValueError(f"Model output sizes are not the same: {[m.output_size for m in models]}")
        self._models = models
        self._lw = _log_softmax(lw)  # mix.rs:69

    @staticmethod
    def init(n_models: int, state_size: int, dataset: Dataset, seed: Optional[int] = None) -> "PPCAMix":
        """mix.rs:76-83"""
        ss = np.random.SeedSequence(seed)
        seeds = [int(s.generate_state(1)[0]) for s in ss.spawn(n_models)]
        return PPCAMix([PPCAModel.init(state_size, dataset, seed=s) for s in seeds], np.zeros(n_models))

    @property
    def output_size(self) -> int:
        return self._models[0].output_size

    @property
    def state_sizes(self) -> List[int]:
        return [m.state_size for m in self._models]

    @property
    def n_parameters(self) -> int:
        """mix.rs:96-104"""
        return sum(m.n_parameters for m in self._models) + len(self._models) - 1

    @property
    def models(self) -> List[PPCAModel]:
        return list(self._models)

    @property
    def log_weights(self) -> np.ndarray:
        return self._lw.copy()

    @property
    def weights(self) -> np.ndarray:
        return np.exp(self._lw)

    def _handles(self, ctx):
        devs = [m._device(ctx) for m in self._models]
        arr = (C.c_void_p * len(devs))(*[d.h for d in devs])
        return devs, arr

    def llk(self, dataset: Dataset) -> float:
        """mix.rs:162-174"""
        devs, arr = self._handles(dataset._ctx)
        tot = C.c_double(0.0)
        check(lib().ppca_mix_llk(dataset._ctx.handle, dataset._h, arr, ptr(self._lw), len(devs), C.byref(tot), None, None))
        return tot.value

    def llks(self, dataset: Dataset) -> np.ndarray:
        """mix.rs:152-159"""
        devs, arr = self._handles(dataset._ctx)
        out = np.empty(len(dataset))
        check(lib().ppca_mix_llk(dataset._ctx.handle, dataset._h, arr, ptr(self._lw), len(devs), None, ptr(out), None))
        return out

    def infer_cluster(self, dataset: Dataset) -> np.ndarray:
        """Log posteriors (N, n_models) (mix.rs:179-189)."""
        devs, arr = self._handles(dataset._ctx)
        out = np.empty((len(dataset), len(devs)))
        check(lib().ppca_mix_llk(dataset._ctx.handle, dataset._h, arr, ptr(self._lw), len(devs), None, None, ptr(out)))
        return out

    def _iterate(self, dataset: Dataset, prior: Optional[Prior], want_llk: bool):
        ctx = dataset._ctx
        if len(dataset) == 0:
            raise ValueError("dataset is empty")
        devs, arr = self._handles(ctx)
        outs = []
        for m in self._models:
            h = C.c_void_p()
            check(lib().ppca_model_alloc(ctx.handle, m.output_size, m.state_size, C.byref(h)))
            outs.append(_DevModel(h))
        oarr = (C.c_void_p * len(outs))(*[o.h for o in outs])
        lw_out = np.empty(len(outs))
        llk = C.c_double(0.0)
        pref, keep = _prior_ref(prior)
        check(lib().ppca_mix_em_step(ctx.handle, dataset._h, arr, ptr(self._lw), len(devs), pref, oarr, ptr(lw_out),
                                     C.byref(llk) if want_llk else None))
        models = [PPCAModel._from_device(o, ctx, m.output_size, m.state_size) for o, m in zip(outs, self._models)]
        new = PPCAMix.__new__(PPCAMix)
        new._models, new._lw = models, lw_out
        return new, (llk.value if want_llk else None)

    def iterate(self, dataset: Dataset) -> "PPCAMix":
        return self._iterate(dataset, None, False)[0]

    def iterate_with_prior(self, dataset: Dataset, prior: Prior) -> "PPCAMix":
        """mix.rs:281-337"""
        return self._iterate(dataset, prior, False)[0]

    def iterate_with_llk(self, dataset: Dataset, prior: Optional[Prior] = None):
        return self._iterate(dataset, prior, True)

    def to_canonical(self) -> "PPCAMix":
        """mix.rs:340-346"""
        new = PPCAMix.__new__(PPCAMix)
        new._models, new._lw = [m.to_canonical() for m in self._models], self._lw.copy()
        return new

    # -- inference outputs (mix.rs:179-265; src/python_bindings.rs:645-672) ---------------------------
    def infer(self, dataset: Dataset) -> "InferredMaskedMix":
        """Posterior over the components and the per-component state posteriors (mix.rs:206-235)."""
        return InferredMaskedMix(self, self.infer_cluster(dataset), [m.infer(dataset) for m in self._models])

    def _mix_recon(self, dataset: Dataset, mode: int) -> Dataset:
        ctx = dataset._ctx
        devs, arr = self._handles(ctx)
        h = C.c_void_p()
        check(lib().ppca_mix_reconstruct(ctx.handle, dataset._h, arr, ptr(self._lw), len(devs), mode, C.byref(h)))
        return Dataset._wrap(h, ctx)

    def smooth(self, dataset: Dataset) -> Dataset:
        """Posterior-weighted sum of the components' smoothed outputs, on the GPU (mix.rs:238-251, :404-412);
        the result carries no weights, like the reference's."""
        return self._mix_recon(dataset, 0)

    filter_extrapolate = smooth

    def extrapolate(self, dataset: Dataset) -> Dataset:
        """mix.rs:254-265, :414-423"""
        return self._mix_recon(dataset, 1)

    def sample(self, dataset_size: int, mask_prob: float, seed: Optional[int] = None) -> Dataset:
        """mix.rs:124-134: a component per sample from the prior weights, then that component's generative
        process (component blocks are generated on the GPU and interleaved by a random permutation)."""
        if not (0.0 <= mask_prob <= 1.0):
            raise ValueError("invalid mask probability")
        rng = np.random.default_rng(seed)
        which = rng.choice(len(self._models), size=dataset_size, p=self.weights / self.weights.sum())
        out = np.empty((dataset_size, self.output_size))
        for c, m in enumerate(self._models):
            idx = np.nonzero(which == c)[0]
            if idx.size:
                out[idx] = m.sample(idx.size, mask_prob, seed=int(rng.integers(0, 2 ** 63 - 1))).numpy()
        return Dataset(out)

    # -- serialisation (own container; bincode layout is a "next" row) ---------------------------------
    def dump(self, format: str = "npz") -> bytes:
        if format == "bincode":
            return _wire.dump_mix([(m._sigma, m._c, m._mean) for m in self._models], self._lw)
        buf = io.BytesIO()
        np.savez(buf, kind="ppca_rs_amd.PPCAMix", log_weights=self._lw, n_models=len(self._models),
                 **{f"sigma_{i}": m._sigma for i, m in enumerate(self._models)},
                 **{f"transform_{i}": m._c for i, m in enumerate(self._models)},
                 **{f"mean_{i}": m._mean for i, m in enumerate(self._models)})
        return buf.getvalue()

    @staticmethod
    def load(data: bytes) -> "PPCAMix":
        try:
            if bytes(data[:2]) != b"PK":
                models, lw = _wire.load_mix(data)
                return PPCAMix([PPCAModel(*m) for m in models], lw)
            z = np.load(io.BytesIO(data), allow_pickle=False)
            nm = int(z["n_models"])
            return PPCAMix([PPCAModel(float(z[f"sigma_{i}"]), z[f"transform_{i}"], z[f"mean_{i}"]) for i in range(nm)],
                           z["log_weights"])
        except Exception as err:
            raise Exception(str(err))

    def __getstate__(self):
        return self.dump()

    def __setstate__(self, state):
        o = PPCAMix.load(state)
        self.__dict__.update(o.__dict__)

    def __getnewargs__(self):
        return (self.models, self.log_weights)


class InferredMaskedMix:
    """Batch of mixture posteriors (mix.rs:349-515; src/python_bindings.rs:713-885).  The per-sample inference
    ran on the GPU (PPCAMix.infer); these accessors combine the already-inferred host arrays exactly as the
    reference's do, quirks included."""

    def __init__(self, mix: PPCAMix, log_posterior: np.ndarray, inferred: List["InferredMasked"]):
        self._mix, self._lp, self._inf = mix, log_posterior, inferred

    def log_posteriors(self) -> np.ndarray:
        if self._lp.shape[0] == 0:
            return np.zeros((0, 0))
        return self._lp.copy()

    def posteriors(self) -> np.ndarray:
        if self._lp.shape[0] == 0:
            return np.zeros((0, 0))
        return np.exp(self._lp)

    def sub_states(self) -> List["InferredMasked"]:
        return list(self._inf)

    def _state_stack(self) -> np.ndarray:  # (K, N, k)
        return np.stack([i._states for i in self._inf])

    def states(self) -> np.ndarray:
        """mix.rs:374-380 -- as written upstream the components are weighted by the LOG posterior."""
        if self._lp.shape[0] == 0:
            return np.zeros((0, 0))
        return np.einsum("nc,cnk->nk", self._lp, self._state_stack())

    def covariances(self) -> List[np.ndarray]:
        """mix.rs:383-396: sum_c post_c (Sigma_c + (z_c - mean)(z_c - mean)^T), mean = states()."""
        mean, post, zs = self.states(), np.exp(self._lp), self._state_stack()
        out = []
        for i in range(self._lp.shape[0]):
            acc = 0.0
            for c, inf in enumerate(self._inf):
                dv = zs[c, i] - mean[i]
                acc = acc + post[i, c] * (inf._covs[i] + np.outer(dv, dv))
            out.append(acc)
        return out

    def _weighted(self, per_model: List[np.ndarray]) -> np.ndarray:
        return np.einsum("nc,cnj->nj", np.exp(self._lp), np.stack(per_model))

    def smoothed(self, ppca: PPCAMix) -> Dataset:
        """mix.rs:399-407"""
        return Dataset(np.ascontiguousarray(self._weighted([i.smoothed(m).numpy() for i, m in zip(self._inf, ppca._models)])))

    def extrapolated(self, ppca: PPCAMix, dataset: Dataset) -> Dataset:
        """mix.rs:410-418"""
        return Dataset(np.ascontiguousarray(
            self._weighted([i.extrapolated(m, dataset).numpy() for i, m in zip(self._inf, ppca._models)])))

    def _cov_sum(self, means: List[np.ndarray], covs: List[List[np.ndarray]]) -> List[np.ndarray]:
        post = np.exp(self._lp)
        mean = np.einsum("nc,cnj->nj", post, np.stack(means))
        out = []
        for i in range(self._lp.shape[0]):
            acc = 0.0
            for c in range(len(self._inf)):
                dv = means[c][i] - mean[i]
                acc = acc + post[i, c] * (covs[c][i] + np.outer(dv, dv))
            out.append(acc)
        return out

    def smoothed_covariances(self, ppca: PPCAMix) -> List[np.ndarray]:
        """mix.rs:426-440 -- d x d per sample."""
        return self._cov_sum([i.smoothed(m).numpy() for i, m in zip(self._inf, ppca._models)],
                             [i.smoothed_covariances(m) for i, m in zip(self._inf, ppca._models)])

    def smoothed_covariances_diagonal(self, ppca: PPCAMix) -> Dataset:
        """mix.rs:447-461"""
        sm = [i.smoothed(m).numpy() for i, m in zip(self._inf, ppca._models)]
        mean = self._weighted(sm)
        dg = [i.smoothed_covariances_diagonal(m).numpy() + (s - mean) ** 2 for i, m, s in zip(self._inf, ppca._models, sm)]
        return Dataset(np.ascontiguousarray(self._weighted(dg)))

    def extrapolated_covariances(self, ppca: PPCAMix, dataset: Dataset) -> List[np.ndarray]:
        """mix.rs:464-477 -- as written upstream each component contributes its SMOOTHED covariance."""
        return self._cov_sum([i.extrapolated(m, dataset).numpy() for i, m in zip(self._inf, ppca._models)],
                             [i.smoothed_covariances(m) for i, m in zip(self._inf, ppca._models)])

    def extrapolated_covariances_diagonal(self, ppca: PPCAMix, dataset: Dataset) -> Dataset:
        """mix.rs:485-505"""
        ex = [i.extrapolated(m, dataset).numpy() for i, m in zip(self._inf, ppca._models)]
        mean = self._weighted(ex)
        dg = [i.extrapolated_covariances_diagonal(m, dataset).numpy() + (e - mean) ** 2
              for i, m, e in zip(self._inf, ppca._models, ex)]
        return Dataset(np.ascontiguousarray(self._weighted(dg)))

    def posterior_sampler(self) -> "PosteriorSamplerMix":
        """mix.rs:508-518"""
        return PosteriorSamplerMix(np.exp(self._lp), [i.posterior_sampler() for i in self._inf])


class PosteriorSamplerMix:
    """mix.rs:521-537; src/python_bindings.rs:887-905: a component per sample from its posterior, then a draw
    from that component's state posterior pushed through its model."""

    def __init__(self, posteriors: np.ndarray, samplers: List["PosteriorSampler"]):
        self._post, self._samplers = posteriors, samplers

    def sample(self, seed: Optional[int] = None) -> Dataset:
        rng = np.random.default_rng(seed)
        n = self._post.shape[0]
        p = self._post / self._post.sum(axis=1, keepdims=True)
        which = (rng.random((n, 1)) > np.cumsum(p, axis=1)).sum(axis=1).clip(0, p.shape[1] - 1)
        draws = [s.sample(seed=int(rng.integers(0, 2 ** 63 - 1))).numpy() for s in self._samplers]
        return Dataset(np.ascontiguousarray(np.stack(draws)[which, np.arange(n)]))


@dataclass
class PPCAMixTrainer:
    """python/ppca_rs/__init__.py:70-118"""

    dataset: Dataset

    def train(self, *, start: Optional[PPCAMix] = None, prior: Optional[Prior] = None, n_models: int,
              state_size: int, n_iters: int = 10, metric: Literal["aic", "bic", "llk"] = "aic", quiet: bool = False,
              seed: Optional[int] = None) -> PPCAMix:
        model = start or PPCAMix.init(n_models, state_size, self.dataset, seed=seed)
        n = len(self.dataset)
        for idx in range(n_iters):
            if not quiet:
                new_model, llk = model.iterate_with_llk(self.dataset, prior)
                metrics = _metrics(llk, model.n_parameters, n)
                print(f"Masked PPCA mix iteration {idx + 1}: {metric}={getattr(metrics, metric)}")
                model = new_model
            else:
                model = model.iterate_with_prior(self.dataset, prior) if prior is not None else model.iterate(self.dataset)
        return model.to_canonical()
