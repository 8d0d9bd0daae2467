"""Long-format DataFrame <-> Dataset adapters.

Same names, fields and semantics as the reference's `DataFrameAdapter` / `DataFrameAdapterDescription`
(python/ppca_rs/__init__.py:119-433): a sample is one value of the `keys` columns, an output dimension one value
of the `dimensions` columns, the `metric` column fills the (sample, dimension) matrix and everything absent stays
masked (NaN).  Host-side convenience only -- the resulting `Dataset` lives on the GPU like any other.  The pivot
itself is vectorised here (factorised group codes + one fancy assignment) instead of a Python loop over groups.
pandas is duck-typed and imported lazily; polars is optional (absent from this image: that branch is untested).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import numpy as np

from .api import Dataset

DIM = "__dim_idx"
SAMPLE = "__sample_idx"


def pivot_pandas(df, *, keys: List[str], dimensions: Optional[List[str]], dimension_idx, metric: str):
    """(matrix (n_samples, n_dims) with NaN where absent, dimension_idx frame, sample_idx frame, dimensions).
    Dimension order: sorted unique dimension tuples (__init__.py:160-168); sample order: sorted key tuples (the
    order of pandas' groupby, :176); rows whose dimensions are not in `dimension_idx` are dropped (inner join,
    :175); a later duplicate of a (sample, dimension) pair overwrites an earlier one (:186-187)."""
    if dimension_idx is None:
        dimension_idx = df[dimensions].drop_duplicates().sort_values(dimensions).reset_index(drop=True)
        dimension_idx.insert(0, DIM, np.arange(len(dimension_idx), dtype=np.int64))
    elif dimensions is None:
        dimensions = [c for c in dimension_idx.columns if c != DIM]
    joined = df.merge(dimension_idx, on=dimensions)
    codes = joined.groupby(keys, sort=True).ngroup().to_numpy()
    n_samples = int(codes.max()) + 1 if len(codes) else 0
    matrix = np.full((n_samples, len(dimension_idx)), np.nan)
    matrix[codes, joined[DIM].to_numpy()] = joined[metric].to_numpy(dtype=np.float64)
    sample_idx = joined[keys].drop_duplicates().sort_values(keys).reset_index(drop=True)
    sample_idx[SAMPLE] = np.arange(len(sample_idx), dtype=np.int64)
    return matrix, dimension_idx, sample_idx[[*keys, SAMPLE]], list(dimensions)


@dataclass
class DataFrameAdapter:
    """python/ppca_rs/__init__.py:119-143"""

    keys: List[str]
    dimensions: List[str]
    metric: str
    dimension_idx: Any
    sample_idx: Any
    dataset: Dataset
    origin: str

    @classmethod
    def from_pandas(cls, df, *, keys: List[str], dimensions: Optional[List[str]] = None, dimension_idx=None,
                    metric: str) -> "DataFrameAdapter":
        """__init__.py:145-206"""
        matrix, dimension_idx, sample_idx, dimensions = pivot_pandas(
            df, keys=keys, dimensions=dimensions, dimension_idx=dimension_idx, metric=metric)
        return cls(keys, dimensions, metric, dimension_idx, sample_idx, Dataset(matrix), origin="pandas")

    @classmethod
    def from_polars(cls, df, *, keys: List[str], dimensions: Optional[List[str]] = None, dimension_idx=None,
                    metric: str) -> "DataFrameAdapter":
        """__init__.py:208-270 (goes through pandas for the pivot; the index frames are handed back as polars)."""
        import polars as pl

        dpd = dimension_idx.to_pandas() if dimension_idx is not None else None
        matrix, dim_pd, smp_pd, dimensions = pivot_pandas(
            df.to_pandas(), keys=keys, dimensions=dimensions, dimension_idx=dpd, metric=metric)
        return cls(keys, dimensions, metric, pl.from_pandas(dim_pd), pl.from_pandas(smp_pd), Dataset(matrix),
                   origin="polars")

    def _dim_pandas(self):
        return self.dimension_idx if self.origin == "pandas" else self.dimension_idx.to_pandas()

    def description(self) -> "DataFrameAdapterDescription":
        """__init__.py:272-299"""
        if self.origin not in ("pandas", "polars"):
            raise Exception(f"Unknown origin {self.origin}")
        dims = self._dim_pandas().sort_values(DIM)
        return DataFrameAdapterDescription(
            keys=self.keys, dimensions=self.dimensions, metric=self.metric,
            dimension_idx=[[row[c] for c in self.dimensions] for row in dims.to_dict("records")])

    def convert_dataset(self, dataset: Dataset, *, column_name: str):
        return self.convert_datasets({column_name: dataset})

    def convert_datasets(self, datasets: Dict[str, Dataset]):
        """Back to long format: one row per (sample, dimension), columns keys + dimensions + one per dataset
        (__init__.py:304-366)."""
        import pandas as pd

        if self.origin not in ("pandas", "polars"):
            raise Exception(f"Unknown origin {self.origin}")
        n_s, n_d = len(self.sample_idx), len(self.dimension_idx)
        dims = self._dim_pandas().sort_values(DIM)
        smps = (self.sample_idx if self.origin == "pandas" else self.sample_idx.to_pandas()).sort_values(SAMPLE)
        out = {k: np.repeat(smps[k].to_numpy(), n_d) for k in self.keys}
        out.update({c: np.tile(dims[c].to_numpy(), n_s) for c in self.dimensions})
        for name, ds in datasets.items():
            arr = ds.numpy()
            if arr.shape != (n_s, n_d):
                raise ValueError(f"dataset {name!r} has shape {arr.shape}, adapter expects {(n_s, n_d)}")
            out[name] = arr.reshape(-1)
        frame = pd.DataFrame(out)
        if self.origin == "polars":
            import polars as pl

            return pl.from_pandas(frame)
        return frame


@dataclass
class DataFrameAdapterDescription:
    """Serialisable recipe of an adapter (python/ppca_rs/__init__.py:369-433)."""

    keys: List[str]
    dimensions: List[str]
    metric: str
    dimension_idx: List[List]

    def _columns(self) -> dict:
        cols = {DIM: np.arange(len(self.dimension_idx), dtype=np.int64)}
        cols.update({dim: [item[i] for item in self.dimension_idx] for i, dim in enumerate(self.dimensions)})
        return cols

    @property
    def dimension_idx_pandas(self) -> Any:
        import pandas as pd

        return pd.DataFrame(self._columns())

    @property
    def dimension_idx_polars(self) -> Any:
        import polars as pl

        return pl.DataFrame(self._columns())

    @classmethod
    def from_json(cls, value: dict) -> "DataFrameAdapterDescription":
        return cls(**value)

    def to_json(self) -> dict:
        return {"keys": self.keys, "dimensions": self.dimensions, "metric": self.metric,
                "dimension_idx": self.dimension_idx}

    def adapt_pandas(self, df) -> DataFrameAdapter:
        return DataFrameAdapter.from_pandas(df, keys=self.keys, dimension_idx=self.dimension_idx_pandas, metric=self.metric)

    def adapt_polars(self, df) -> DataFrameAdapter:
        return DataFrameAdapter.from_polars(df, keys=self.keys, dimension_idx=self.dimension_idx_polars, metric=self.metric)
