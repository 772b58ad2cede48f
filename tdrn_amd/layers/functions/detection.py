"""Detect: drop-in for layers/functions/detection.py:8-70.  The whole post-process (two-stage
decode, per-class score threshold, cpu_nms-exact greedy NMS, top-k pack) runs on the device in
three launches of libtdrn_hip.so instead of B x (C-1) python iterations with PCIe round trips."""
import ctypes as C

import torch

from ... import _lib


class Detect(object):
    def __init__(self, num_classes, bkg_label, top_k, conf_thresh, nms_thresh):
        self.num_classes = num_classes
        self.background_label = bkg_label
        self.top_k = top_k
        self.nms_thresh = nms_thresh
        if nms_thresh <= 0:
            raise ValueError('nms_threshold must be non negative.')
        self.conf_thresh = conf_thresh
        self.variance = [0.1, 0.2]
        self._ws = None
        self.last_counts = None

    def forward(self, loc_data, conf_data, prior_data, arm_loc_data=None, scale=None, feature=None, out=None):
        """loc (B,P,4), conf (B*P,C), priors (P,4), arm_loc (B,P,4)|None, scale 4-vector (default
        [320]*4 like detection.py:25).  `feature` is accepted and ignored (test_video.py:115 passes
        it).  Returns (B, C, top_k, 5) rows [score, x1, y1, x2, y2] on the inputs' device.
        `out` (not in the reference): a preallocated (B, C, top_k, 5) fp32 tensor the rows are written into -- on the device, or in
        PINNED host memory (the kernel's only use of it is one coalesced write per (image, class) row, so the detections then cross
        PCIe as part of the launch and no device-to-host copy follows: tdrn_amd/stream.py)."""
        _lib.require_cuda(loc_data, "loc_data")
        dev = loc_data.device
        B, P, Cn = loc_data.size(0), prior_data.size(0), self.num_classes
        loc = loc_data.contiguous().float()
        conf = conf_data.contiguous().float().view(-1, Cn)
        if conf.size(0) != B * P:
            raise ValueError("conf_data has %d rows, expected %d" % (conf.size(0), B * P))
        pri = prior_data.to(dev).contiguous().float()
        arm = arm_loc_data.contiguous().float() if arm_loc_data is not None else None
        lib = _lib.lib()
        # evaluate.py:461 passes a CUDA tensor: it is read on the device (no .tolist() = no hidden device sync)
        scale_d = None
        if isinstance(scale, torch.Tensor) and scale.is_cuda:
            scale_d = scale.detach().to(dev, torch.float32).contiguous().view(-1)
            if scale_d.numel() != 4:
                raise ValueError("scale must have 4 elements")
            if scale_d.data_ptr() % 16:
                scale_d = scale_d.clone()
        else:
            sc = [320.0] * 4 if scale is None else [float(v) for v in (scale.tolist() if hasattr(scale, "tolist") else scale)]
            scale_h = (C.c_float * 4)(*sc)
        nb = lib.tdrn_detect_workspace_bytes(B, P, Cn, self.top_k)
        if self._ws is None or self._ws.numel() < nb or self._ws.device != dev:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        if out is None:
            out = torch.empty((B, Cn, self.top_k, 5), dtype=torch.float32, device=dev)
        else:
            if tuple(out.shape) != (B, Cn, self.top_k, 5) or out.dtype != torch.float32 or not out.is_contiguous():
                raise ValueError("out must be a contiguous fp32 tensor of shape %s" % ((B, Cn, self.top_k, 5),))
            if not (out.is_cuda and out.device == dev) and not (out.device.type == "cpu" and out.is_pinned()):
                raise ValueError("out must live on %s or in pinned host memory" % (dev,))
        counts = torch.empty((B, Cn), dtype=torch.int32, device=dev)
        fn, sarg = (lib.tdrn_detect_dev_scale, _lib.ptr(scale_d)) if scale_d is not None else (lib.tdrn_detect, scale_h)
        _lib.check(fn(_lib.ptr(loc), _lib.ptr(conf), _lib.ptr(pri), _lib.ptr(arm), sarg, B, P, Cn,
                      self.top_k, float(self.conf_thresh), float(self.nms_thresh), _lib.ptr(out),
                      _lib.ptr(counts), _lib.ptr(self._ws), self._ws.numel(),
                      _lib.current_stream(dev)), "tdrn_detect")
        self.last_counts = counts
        return out

    __call__ = forward
