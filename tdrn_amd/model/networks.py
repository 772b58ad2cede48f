"""Parameter-holder builders mirroring model/networks.py of the reference (vgg :136-163,
conv_dw :736-745) and the deformable-conv module surface (conv_offset2d :600-615,
ConvOffset2d :699-733).  The modules only own parameters with the reference's names and
shapes; arithmetic happens in libtdrn_hip.so."""
import ctypes as C
import math

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from .. import _lib

vgg_base = {k: [64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512]
            for k in ("300", "320", "512")}


def vgg(cfg, i, batch_norm=False, pool5_ds=False, c7_channel=1024):
    layers, cin = [], i
    for v in cfg:
        if v in ("M", "C"):
            layers.append(nn.MaxPool2d(2, 2, ceil_mode=(v == "C")))
            continue
        layers.append(nn.Conv2d(cin, v, 3, padding=1))
        if batch_norm:
            layers.append(nn.BatchNorm2d(v))
        layers.append(nn.ReLU(inplace=True))
        cin = v
    layers.append(nn.MaxPool2d(2, 2) if pool5_ds else nn.MaxPool2d(3, 1, 1))
    for conv in (nn.Conv2d(512, 1024, 3, padding=6, dilation=6), nn.Conv2d(1024, c7_channel, 1)):
        layers.append(conv)
        if batch_norm:
            layers.append(nn.BatchNorm2d(conv.out_channels))
        layers.append(nn.ReLU(inplace=True))
    return layers


def conv_dw(inp, oup, stride):
    return nn.Sequential(nn.Conv2d(inp, inp, 3, stride, 1, groups=inp, bias=False), nn.BatchNorm2d(inp),
                         nn.ReLU(inplace=True),
                         nn.Conv2d(inp, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup), nn.ReLU(inplace=True))


def mobilenet_backbone(c_last=1024):
    plan = [(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 1), (256, 256, 1), (256, 512, 2)] + \
           [(512, 512, 1)] * 5 + [(512, 1024, 2), (1024, c_last, 1)]
    first = nn.Sequential(nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True))
    return nn.ModuleList([first] + [conv_dw(a, b, s) for a, b, s in plan])


def conv_offset2d(input, offset, weight, stride=1, padding=0, dilation=1, deform_groups=1, compute="fp32"):
    """Deformable conv v1 forward through tdrn_deform_conv_forward (replaces the FFI call at
    model/networks.py:641-645).  NCHW fp32 CUDA tensors in, NCHW fp32 out, no bias."""
    if input is not None and input.dim() != 4:
        raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
    _lib.require_cuda(input, "input")
    lib = _lib.lib()
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    x = input.contiguous().float()
    off = offset.contiguous().float()
    w = weight.detach().contiguous().float()
    N, Cin, H, W = x.shape
    Cout, Cw, kh, kw = w.shape
    if Cw != Cin:
        raise RuntimeError("invalid number of input planes, expected: %d, but got: %d" % (Cw, Cin))
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if Ho < 1 or Wo < 1:
        raise ValueError("convolution input is too small (output would be {}x{})".format(Ho, Wo))
    if tuple(off.shape) != (N, deform_groups * 2 * kh * kw, Ho, Wo):
        raise RuntimeError("invalid shape of offset: expected %r, got %r"
                           % ((N, deform_groups * 2 * kh * kw, Ho, Wo), tuple(off.shape)))
    dt = _lib.DTYPES[compute]
    nb = lib.tdrn_deform_conv_workspace_bytes(N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, deform_groups, dt)
    if nb == 0:
        raise RuntimeError("deform_conv: shape check failed")
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    out = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    _lib.check(lib.tdrn_deform_conv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(off), _lib.ptr(out), N, Cin, H, W, Cout,
                                            kw, kh, sw, sh, pw, ph, dh, dw, deform_groups, dt, _lib.ptr(ws), nb,
                                            _lib.current_stream(x.device)), "tdrn_deform_conv_forward")
    return out


class ConvOffset2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 num_deformable_groups=1):
        super(ConvOffset2d, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.num_deformable_groups = num_deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, input, offset):
        return conv_offset2d(input, offset, self.weight, self.stride, self.padding, self.dilation,
                             self.num_deformable_groups)
