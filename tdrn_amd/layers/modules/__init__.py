from .l2norm import L2Norm

__all__ = ['L2Norm']
