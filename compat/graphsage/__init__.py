"""`graphsage` import surface of the reference, served by graphsage_amd.

Put this directory on PYTHONPATH (next to the repository root) and the reference's own scripts
import what they expect — `from graphsage import settings`, `from graphsage.layers import SAGE`,
`from graphsage.samplers import UniformLoader`, `from graphsage.models import GraphSAGE`,
`from graphsage.trainers import SupervisedGraphSageTrainerForNodeLevelTask`,
`from graphsage.datasets import Planetoid` — from the MI355X implementation.  The reference is
also run with its package directory itself on the path (`from models.graphsage import GraphSAGE`,
`from samplers import ...`, `from datasets import ...`: examples/reddit_graphsage_supervised.py:9,
trainers/base_trainers.py:15); the sibling packages models / samplers / datasets / layers here
serve those spellings.  Nothing in here is code of the reference: every name is re-exported
from graphsage_amd."""
import importlib
import sys

import graphsage_amd as _impl

_SUBMODULES = (
    "settings", "data", "layers", "layers.sage", "samplers", "samplers.uniform_loader", "samplers.utils",
    "models", "models.graphsage", "datasets", "trainers", "trainers.base_trainers",
    "trainers.node_level_graphsage_trainers", "trainers.graph_level_graphsage_trainers", "experiments",
)
for _name in _SUBMODULES:
    _mod = importlib.import_module(f"graphsage_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    if "." not in _name:
        globals()[_name] = _mod
__version__ = getattr(_impl, "__version__", "0")
