"""Prior-box configs and the frame preprocess of the reference (data/__init__.py:7-23), on the device."""
import ctypes as C

import torch

from .. import _lib
from .config import mb_cfg, multi_cfg, multi_cfg_512, VOC_320, VOC_512_RefineDet   # noqa: F401

MEANS = (104, 117, 123)   # BGR order, like every driver of the reference (evaluate.py:83, test_video.py:36)


def base_transform(image, size, mean, to_rgb=False):
    """uint8 BGR frame(s) on the GPU, (H,W,3) or (B,H,W,3) -> float32 (B,3,size,size): cv2-style bilinear resize,
    minus `mean` (BGR order), optional BGR->RGB (data/voc0712.py:467-468 swaps, test_video.py:103-105 does not),
    HWC->CHW.  One kernel (tdrn_preprocess); the result feeds net(x) directly."""
    _lib.require_cuda(image, "image")
    if image.dtype != torch.uint8:
        raise TypeError("base_transform expects uint8 frames")
    x = image.unsqueeze(0) if image.dim() == 3 else image
    x = x.contiguous()
    B, H0, W0, ch = x.shape
    if ch != 3:
        raise ValueError("expected BGR frames (..., 3)")
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=x.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    _lib.check(_lib.lib().tdrn_preprocess(_lib.ptr(x), B, H0, W0, size, m, int(bool(to_rgb)), _lib.ptr(out),
                                          _lib.current_stream(x.device)), "tdrn_preprocess")
    return out


class BaseTransform(object):
    def __init__(self, size, mean, to_rgb=False):
        self.size, self.mean, self.to_rgb = size, tuple(float(v) for v in mean), to_rgb

    def __call__(self, image, boxes=None, labels=None):
        return base_transform(image, self.size, self.mean, self.to_rgb), boxes, labels
