from .graphsage import GraphSAGE

__all__ = ["GraphSAGE"]
