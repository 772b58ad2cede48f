"""nms(dets, thresh, force_cpu): drop-in for utils/nms_wrapper.py:23-31.  Both branches run on the
device; `force_cpu=True` keeps cpu_nms's rule (suppress on IoU >= thresh, utils/nms/cpu_nms.pyx:66),
`force_cpu=False` keeps gpu_nms's (host argsort, strict >, utils/nms/gpu_nms.pyx:16-31)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def cpu_nms(dets, thresh):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    lib = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.from_numpy(dets).to(dev)
    keep = torch.empty(n, dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = lib.tdrn_nms_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.tdrn_nms(_lib.ptr(d), n, float(thresh), 0, _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), nb,
                            _lib.current_stream(dev)), "tdrn_nms")
    return keep[: int(num.item())].cpu().tolist()


def gpu_nms(dets, thresh, device_id=None):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    if n == 0:
        return []
    if device_id is None:          # the reference defaults to 0 (single-GPU scripts); a sharded rank uses its own GPU
        device_id = torch.cuda.current_device()
    order = dets[:, 4].argsort()[::-1]
    sorted_dets = np.ascontiguousarray(dets[order, :])
    keep = np.zeros(n, dtype=np.int32)
    num = C.c_int(0)
    _lib.check(_lib.lib().tdrn_gpu_nms_host(keep.ctypes.data_as(C.c_void_p), C.byref(num),
                                            sorted_dets.ctypes.data_as(C.c_void_p), n, dets.shape[1],
                                            float(thresh), device_id), "tdrn_gpu_nms_host")
    return list(order[keep[: num.value]])


def nms(dets, thresh, force_cpu=False):
    if dets.shape[0] == 0:
        return []
    if force_cpu:
        return cpu_nms(dets, thresh)
    return gpu_nms(dets, thresh)
