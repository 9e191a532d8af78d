from .base import BaseVictim
from .lightgcn import LightGCN
from .mf import MF

__all__ = ["BaseVictim", "LightGCN", "MF"]
