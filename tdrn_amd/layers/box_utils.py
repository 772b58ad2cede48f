"""decode / center_size on the device (layers/box_utils.py:176-195, :16-25 of the reference)."""
import torch

from .. import _lib


def decode(loc, priors, variances):
    _lib.require_cuda(loc, "loc")
    loc = loc.contiguous().float()
    pri = priors.to(loc.device).contiguous().float()
    out = torch.empty_like(loc)
    _lib.check(_lib.lib().tdrn_decode(_lib.ptr(loc), _lib.ptr(pri), loc.size(0), float(variances[0]),
                                      float(variances[1]), _lib.ptr(out), _lib.current_stream(loc.device)))
    return out


def center_size(boxes):
    _lib.require_cuda(boxes, "boxes")
    boxes = boxes.contiguous().float()
    out = torch.empty_like(boxes)
    _lib.check(_lib.lib().tdrn_center_size(_lib.ptr(boxes), boxes.size(0), _lib.ptr(out),
                                           _lib.current_stream(boxes.device)))
    return out
