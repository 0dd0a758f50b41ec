"""Top-level `layers` as the reference's scripts spell it when run with graphsage/ on the path
(see compat/graphsage/__init__.py); everything comes from graphsage_amd.layers."""
import importlib
import sys

from graphsage_amd.layers import *  # noqa: F401,F403
from graphsage_amd import layers as _impl

for _sub in ('sage',):
    sys.modules[f"{__name__}.{_sub}"] = importlib.import_module(f"graphsage_amd.layers.{_sub}")
__all__ = getattr(_impl, "__all__", [n for n in dir(_impl) if not n.startswith("_")])
