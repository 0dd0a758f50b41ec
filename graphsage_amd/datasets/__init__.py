"""Synthetic same-shape stand-ins for the PyG datasets the reference re-exports
(reference graphsage/datasets/__init__.py:1-4: Flickr, Reddit, PPI, Planetoid).  There is no
network and no torch_geometric in this image, so the real files cannot be fetched; the
generators reproduce the published shapes with planted-partition structure so
that accuracy/F1 comparisons are meaningful."""
from .synthetic import (Flickr, PPI, Planetoid, Reddit, SyntheticDataset, planted_partition,
                        reddit_like_device)

__all__ = ["Flickr", "PPI", "Planetoid", "Reddit", "SyntheticDataset", "planted_partition",
           "reddit_like_device"]
