"""graphsage_amd — MI355X (gfx950) implementation of GraphSAGE's sampled-neighbourhood
mini-batch hot path behind the module API of wilcoln/graphsage
(graphsage.layers.SAGE, graphsage.samplers.UniformLoader, graphsage.models.GraphSAGE).

The arithmetic lives in libsage_hip.so (hand-written HIP, C ABI in include/sage_hip.h);
this package is the host-side mirror of the reference interface.  There is no CPU path.
"""
from . import _lib  # noqa: F401

__all__ = ["layers", "models", "samplers", "datasets", "settings", "ops", "functional", "data"]
