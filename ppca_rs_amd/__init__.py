"""ppca_rs_amd -- MI355X-native PPCA EM engine with the ppca_rs Python surface.

Drop-in for the EM hot path of viodotcom/ppca_rs (`import ppca_rs_amd as ppca_rs`):
Dataset, Prior, PPCAModel (llk / llks / infer / smooth / extrapolate / iterate /
iterate_with_prior / to_canonical), InferredMasked, PPCAMix and the trainers, over
hand-written HIP kernels for gfx950 behind the C-ABI of include/ppca_hip.h.
"""
from .api import (Dataset, DatasetChunks, InferredMasked, InferredMaskedMix, PosteriorSampler, PosteriorSamplerMix, PPCAMix,
                  PPCAMixTrainer, PPCAModel, PPCATrainer, Prior, TrainMetrics)
from ._lib import PPCAError
from .frames import DataFrameAdapter, DataFrameAdapterDescription

__version__ = "0.1.0"
__all__ = ["Dataset", "DatasetChunks", "InferredMasked", "InferredMaskedMix", "PosteriorSampler", "PosteriorSamplerMix", "PPCAMix", "PPCAMixTrainer", "PPCAModel",
           "PPCATrainer", "Prior", "TrainMetrics", "PPCAError", "DataFrameAdapter", "DataFrameAdapterDescription", "__version__"]
