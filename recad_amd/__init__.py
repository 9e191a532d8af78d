"""recad_amd -- MI355X-native victim-model hot path for RecAD-style workflows.

    from recad_amd import model, dataset, workflow
    victim = model.from_config("victim", "lightgcn", latent_dim_rec=64)

Seeds python/numpy/torch RNGs at import like the reference does (recad/__init__.py:11-14).
"""
import random

import numpy as np
import torch

from . import dataset, default, model, utils, victim, workflow  # noqa: F401
from .default import SEED

random.seed(SEED)
np.random.seed(SEED)
torch.manual_seed(SEED)

__version__ = "0.1.0"
