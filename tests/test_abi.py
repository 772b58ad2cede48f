"""CPU-side checks of the drop-in boundary: libtdrn_hip.so loads and exports every symbol that
include/tdrn_hip.h declares, the host-side pieces (plan builder, PriorBox) behave like the
reference, and the product path refuses to run without a GPU instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tdrn_amd import _lib
from tdrn_amd.data import mb_cfg
from tdrn_amd.engine import NetEngine
from tdrn_amd.layers import Detect, PriorBox

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tdrn_hip.h")).read()
    declared = set(re.findall(r"TDRN_API\s+[\w\s\*]+?\b(tdrn_\w+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libtdrn_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b"gfx950" in _lib.lib().tdrn_version()


def test_no_product_import_of_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "tdrn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/|libtdrn_oracle", txt, re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, "product code must never touch oracle/: %r" % bad


def test_priorbox_host_bit_exact(golden_dir):
    for name in ("VOC_320", "VOC_512_RefineDet"):
        ref = np.load(os.path.join(golden_dir, "priorbox_%s.npz" % name))["priors"]
        got = PriorBox(mb_cfg[name]).forward().numpy()
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(ValueError):
        PriorBox(dict(mb_cfg["VOC_320"], variance=[0.1, -1]))


@pytest.mark.parametrize("modname,kind,kw,nparams", [
    ("dualrefinedet_vggbn", _lib.DRN_VGGBN, dict(multihead=True), 36239936),
    ("dualrefinedet_vggbn", _lib.DRN_VGGBN, dict(multihead=False), None),
    ("dualrefinedet_mobilenet", _lib.DRN_MOBILENET, dict(multihead=True), 18167296),
    ("ssd4scale_mobile", _lib.SSD4SCALE_MOBILE, dict(), 5601388),
    ("ssd4scale_mobile", _lib.SSD4SCALE_MOBILE, dict(deform=True), None),
])
def test_plan_parameter_layout_matches_module(modname, kind, kw, nparams):
    import importlib
    mod = importlib.import_module("tdrn_amd.model." + modname)
    if modname == "dualrefinedet_vggbn":
        net = mod.build_net("test", 320, 21, 1024, 1, True, kw["multihead"])
    elif modname == "dualrefinedet_mobilenet":
        net = mod.build_net("test", 320, 21, 1, kw["multihead"])
    else:
        net = mod.build_net("test", 320, 21, 1024, kw.get("deform", False))
    if nparams:
        assert sum(p.numel() for p in net.parameters()) == nparams      # SURVEY.md 8a / BASELINE.md
    sd = {k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    eng = NetEngine(model=kind, size=320, **kw)
    specs = dict(eng.param_specs())
    assert specs == sd
    assert eng.num_priors == 6375
    assert eng.lib.tdrn_net_workspace_bytes(eng.handle, 2) == 2 * eng.lib.tdrn_net_workspace_bytes(eng.handle, 1)


def test_build_net_rejects_other_sizes(capsys):
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    assert build_net("test", 300) is None                     # dualrefinedet_vggbn.py:218-220
    assert "only SSD320 and SSD512" in capsys.readouterr().out
    h = C.c_void_p()
    cfg = _lib.NetConfig(model=_lib.DRN_VGGBN, size=300, num_classes=21, c7_channel=1024, def_groups=1, bn=1)
    assert _lib.lib().tdrn_net_create(C.byref(cfg), C.byref(h)) == -1
    assert _lib.lib().tdrn_net_create(None, C.byref(h)) == -1


def test_512_plan_and_priors():
    eng = NetEngine(model=_lib.DRN_VGGBN, size=512, multihead=True, dtype="fp16")
    assert eng.num_priors == 16320
    assert PriorBox(mb_cfg["VOC_512_RefineDet"]).forward().shape == (16320, 4)


def test_state_errors_without_gpu():
    eng = NetEngine(model=_lib.SSD4SCALE_MOBILE, size=320)
    lib = eng.lib
    x = np.zeros(4, np.float32)
    assert lib.tdrn_net_set_param(eng.handle, b"no.such.weight", x.ctypes.data_as(C.c_void_p), 4) == -5
    assert lib.tdrn_net_set_param(eng.handle, b"backbone.0.0.weight", x.ctypes.data_as(C.c_void_p), 4) == -5
    io = _lib.NetIO()
    assert lib.tdrn_net_forward(eng.handle, None, None, 0, C.byref(io), None) == -6    # weights not packed
    assert "order" in _lib.error_string(-6)
    # packing without all parameters set is refused before any device work
    assert lib.tdrn_net_pack_weights(eng.handle, C.c_void_p(16), 1 << 30, None) == -5


def test_detect_ctor_and_cpu_refusal():
    with pytest.raises(ValueError):
        Detect(21, 0, 200, 0.01, 0.0)                           # detection.py:20-21
    det = Detect(21, 0, 200, 0.01, 0.45)
    with pytest.raises(NotImplementedError):                    # no CPU fallback
        det.forward(torch.zeros(1, 10, 4), torch.zeros(10, 21), torch.zeros(10, 4))
    from tdrn_amd.model.networks import ConvOffset2d
    m = ConvOffset2d(6, 4, 3, padding=1)
    with pytest.raises(NotImplementedError):                    # same as networks.py:632-633
        m(torch.zeros(1, 6, 5, 5), torch.zeros(1, 18, 5, 5))
    with pytest.raises(ValueError):
        from tdrn_amd.model.networks import conv_offset2d
        conv_offset2d(torch.zeros(6, 5, 5), None, None)


def test_deform_workspace_query_mirrors_shape_check():
    lib = _lib.lib()
    q = lambda *a: lib.tdrn_deform_conv_workspace_bytes(*a)
    assert q(1, 6, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1, 0) > 0
    assert q(1, 6, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1, 4, 0) == 0      # Cin % G != 0
    assert q(1, 6, 2, 2, 4, 3, 3, 1, 1, 0, 0, 1, 1, 1, 0) == 0      # input smaller than kernel
    assert q(1, 6, 8, 8, 4, 0, 3, 1, 1, 1, 1, 1, 1, 1, 0) == 0      # kernel size must be > 0
    assert q(1, 6, 8, 8, 4, 3, 3, 0, 1, 1, 1, 1, 1, 1, 0) == 0      # stride must be > 0
