from .base import BaseVictim
from .lightgcn import LightGCN
from .mf import MF
from .ncf import NCF

__all__ = ["BaseVictim", "LightGCN", "MF", "NCF"]
