"""Trainers with the reference's class names (reference graphsage/trainers/__init__.py)."""
from .base_trainers import (BaseTrainer, GraphSageBaseTrainer, SupervisedGraphSageBaseTrainer,
                            SupervisedTorchModuleBaseTrainer, TorchModuleBaseTrainer, dataloader_kwargs)
from .node_level_graphsage_trainers import (SupervisedGraphSageTrainerForNodeLevelTask,
                                            UnsupervisedGraphSageTrainerForNodeLevelTask)
from .graph_level_graphsage_trainers import (SupervisedGraphSageTrainerForGraphLevelTask,
                                             UnsupervisedGraphSageTrainerForGraphLevelTask)
