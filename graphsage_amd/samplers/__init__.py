from .uniform_loader import NeighborLoader as UniformLoader
from .uniform_loader import NeighborLoader, NeighborSampler
from .utils import get_pos_neg_batches

__all__ = ["UniformLoader", "NeighborLoader", "NeighborSampler", "get_pos_neg_batches"]
