from .detection import Detect
from .prior_box import PriorBox

__all__ = ['Detect', 'PriorBox']
