"""Import shim for the upstream reference (ONLY usable in the build container).

Used by tests/golden/make_golden.py and the `-m "not gpu"` pinning tests to import the
reference's Python modules from /root/reference on CPU without modifying them
(SURVEY.md Appendix B).  Nothing here is shipped to, or importable on, the GPU box:
the product path and the `-m gpu` tests never import this file.
"""
import os
import sys
import types

REF = os.environ.get("TDRN_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "layers"))


def install(cpu_nms=None):
    """Register stub modules, then the reference's packages become importable.

    cpu_nms: callable(dets, thresh) -> keep list, standing in for the un-buildable
    Cython `utils.nms.cpu_nms.cpu_nms` (the oracle's `>=`-threshold restatement).
    Returns a dict of the imported reference entry points.
    """
    import torch

    if REF not in sys.path:
        sys.path.insert(0, REF)
    # our own package mirrors the reference's top-level names; make sure the reference wins here
    for name in ("model", "layers", "data", "utils"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(REF):
            if not (hasattr(mod, "__path__") and any(str(p).startswith(REF) for p in mod.__path__)):
                del sys.modules[name]

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("cv2")
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    stub("utils._ext").deform_conv = stub("utils._ext.deform_conv")
    stub("utils.nms.gpu_nms", gpu_nms=None)
    if cpu_nms is None:
        import importlib
        cpu_nms = importlib.import_module("utils.nms.py_cpu_nms").py_cpu_nms
    stub("utils.nms.cpu_nms", cpu_nms=cpu_nms, cpu_soft_nms=None)
    data = stub("data")
    data.__path__ = [os.path.join(REF, "data")]

    torch.cuda.FloatTensor = torch.FloatTensor
    torch.Tensor.cuda = lambda self, *a, **k: self

    from data.config import mb_cfg
    from layers.functions import Detect, PriorBox
    from layers.box_utils import decode, center_size
    from layers.modules.l2norm import L2Norm
    import model.networks as networks
    return dict(mb_cfg=mb_cfg, Detect=Detect, PriorBox=PriorBox, decode=decode,
                center_size=center_size, L2Norm=L2Norm, networks=networks)
