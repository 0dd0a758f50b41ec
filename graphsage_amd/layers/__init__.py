from .sage import SAGE

__all__ = ["SAGE"]
