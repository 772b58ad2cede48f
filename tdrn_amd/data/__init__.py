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


class U8Frames(object):
    """A batch that stays uint8 until the first conv reads it (SURVEY 8f rank 1 in full): `planes` (B,3,S,S) uint8 on the GPU in the net's
    channel order, `mean` per plane.  net(U8Frames) = net(planes.float() - mean[:, None, None]) bit for bit; no fp32 copy of the batch is
    written when the plan's first conv is computed by conv1_2's producers (16-bit plans of the VGG trunks, tdrn_net_io.reserved[3])."""

    def __init__(self, planes, mean):
        _lib.require_cuda(planes, "planes")
        if planes.dtype != torch.uint8 or planes.dim() != 4 or planes.size(1) != 3:
            raise TypeError("U8Frames expects (B,3,S,S) uint8 planes")
        self.planes = planes.contiguous()
        self.mean = tuple(float(v) for v in mean)
        if len(self.mean) != 3:
            raise ValueError("mean must have 3 entries (plane order)")

    # enough of the tensor surface for the model mirrors (engine_for, forward)
    device = property(lambda self: self.planes.device)
    shape = property(lambda self: self.planes.shape)

    def size(self, i=None):
        return self.planes.size() if i is None else self.planes.size(i)

    def dim(self):
        return 4

    def float(self):
        """the fp32 tensor the reference's BaseTransform hands to the net"""
        return self.planes.float() - torch.tensor(self.mean, dtype=torch.float32, device=self.planes.device).view(1, 3, 1, 1)


def base_transform_u8(image, size, mean, to_rgb=False):
    """base_transform with the frame left uint8: resize only (tdrn_preprocess_u8); the mean -- reordered with the planes when `to_rgb`
    -- travels with the planes and is subtracted inside the net's first conv."""
    _lib.require_cuda(image, "image")
    if image.dtype != torch.uint8:
        raise TypeError("base_transform_u8 expects uint8 frames")
    x = image.unsqueeze(0) if image.dim() == 3 else image
    x = x.contiguous()
    B, H0, W0, ch = x.shape
    if ch != 3:
        raise ValueError("expected BGR frames (..., 3)")
    out = torch.empty((B, 3, size, size), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().tdrn_preprocess_u8(_lib.ptr(x), B, H0, W0, size, int(bool(to_rgb)), _lib.ptr(out),
                                             _lib.current_stream(x.device)), "tdrn_preprocess_u8")
    m = [float(v) for v in mean]
    return U8Frames(out, m[::-1] if to_rgb else m)


class BaseTransform(object):
    def __init__(self, size, mean, to_rgb=False):
        self.size, self.mean, self.to_rgb = size, tuple(float(v) for v in mean), to_rgb

    def __call__(self, image, boxes=None, labels=None):
        return base_transform(image, self.size, self.mean, self.to_rgb), boxes, labels
