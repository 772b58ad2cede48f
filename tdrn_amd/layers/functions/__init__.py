from .detection import Detect
from .detection_ota import Detect as DetectOTA   # layers/functions/__init__.py:3 of the reference
from .prior_box import PriorBox

__all__ = ['Detect', 'DetectOTA', 'PriorBox']
