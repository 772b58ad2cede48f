"""CPU ORACLE -- numpy/ctypes front end of oracle/tdrn_oracle.c (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (tdrn_amd) never does: it fails loudly when libtdrn_hip.so is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtdrn_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "tdrn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_deform_conv_forward.restype = C.c_int
        _lib.orc_prior_box.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def deform_conv_forward(inp, offset, weight, stride=1, padding=0, dilation=1, deform_groups=1):
    """utils/deformconv/deform_conv_cuda.c:98-213.  NCHW fp32 in/out, no bias."""
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    (sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(padding), pair(dilation)
    inp, offset, weight = _f32(inp), _f32(offset), _f32(weight)
    N, Cin, H, W = inp.shape
    Cout, Cw, kh, kw = weight.shape
    if Cw != Cin:
        raise RuntimeError("invalid number of input planes")
    Hc = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wc = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if offset.shape != (N, deform_groups * 2 * kh * kw, Hc, Wc):
        raise RuntimeError("invalid shape of offset: %r" % (offset.shape,))
    out = np.empty((N, Cout, Hc, Wc), np.float32)
    rc = lib().orc_deform_conv_forward(
        inp.ctypes.data_as(C.c_void_p), offset.ctypes.data_as(C.c_void_p),
        weight.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
        N, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, deform_groups)
    if rc != 0:
        raise RuntimeError("orc_deform_conv_forward: shape check failed (%d)" % rc)
    return out


def deform_im2col(im, offset, kh, kw, pad, stride=1, dil=1, G=1):
    """utils/deformconv/deform_conv_cuda_kernel.cu:156-208 for one image (C,H,W)."""
    im, offset = _f32(im), _f32(offset)
    Cc, H, W = im.shape
    Hc = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wc = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    col = np.empty((Cc * kh * kw, Hc * Wc), np.float32)
    lib().orc_deform_im2col(im.ctypes.data_as(C.c_void_p), offset.ctypes.data_as(C.c_void_p),
                            Cc, H, W, kh, kw, pad, pad, stride, stride, dil, dil, G,
                            col.ctypes.data_as(C.c_void_p))
    return col


def cpu_nms(dets, thresh, strict_gt=False):
    """utils/nms/cpu_nms.pyx:17-68 -> list of kept indices (descending score)."""
    dets = _f32(dets)
    n = dets.shape[0]
    if n == 0:
        return []
    keep = np.empty(n, np.int32)
    num = np.zeros(1, np.int32)
    lib().orc_cpu_nms(dets.ctypes.data_as(C.c_void_p), n, C.c_double(float(thresh)),
                      int(bool(strict_gt)), keep.ctypes.data_as(C.c_void_p),
                      num.ctypes.data_as(C.c_void_p))
    return keep[: int(num[0])].tolist()


def decode(loc, priors, variances=(0.1, 0.2)):
    """layers/box_utils.py:176-195."""
    loc, priors = _f32(loc), _f32(priors)
    out = np.empty_like(loc)
    lib().orc_decode(loc.ctypes.data_as(C.c_void_p), priors.ctypes.data_as(C.c_void_p),
                     loc.shape[0], C.c_float(variances[0]), C.c_float(variances[1]),
                     out.ctypes.data_as(C.c_void_p))
    return out


def center_size(boxes):
    """layers/box_utils.py:16-25."""
    boxes = _f32(boxes)
    out = np.empty_like(boxes)
    lib().orc_center_size(boxes.ctypes.data_as(C.c_void_p), boxes.shape[0],
                          out.ctypes.data_as(C.c_void_p))
    return out


def prior_box(cfg):
    """layers/functions/prior_box.py:33-64 -> (P,4) float32."""
    fm = np.asarray(cfg["feature_maps"], np.int32)
    n = len(fm)
    steps = np.asarray(cfg["steps"], np.float64)
    mins = np.asarray(cfg["min_sizes"], np.float64)
    maxs = np.asarray(cfg["max_sizes"], np.float64)
    arc = np.asarray([len(a) for a in cfg["aspect_ratios"]][:n], np.int32)
    ars = np.asarray([v for a in cfg["aspect_ratios"][:n] for v in a], np.float64)
    args = (n, fm.ctypes.data_as(C.c_void_p), C.c_double(cfg["min_dim"]),
            steps.ctypes.data_as(C.c_void_p), mins.ctypes.data_as(C.c_void_p),
            maxs.ctypes.data_as(C.c_void_p), len(maxs), arc.ctypes.data_as(C.c_void_p),
            ars.ctypes.data_as(C.c_void_p), int(cfg["clip"]), int(cfg["flip"]))
    P = lib().orc_prior_box(*args, None)
    out = np.empty((P, 4), np.float32)
    lib().orc_prior_box(*args, out.ctypes.data_as(C.c_void_p))
    return out


def detect(loc, conf, priors, arm_loc=None, scale=(320, 320, 320, 320), num_classes=21,
           top_k=200, conf_thresh=0.01, nms_thresh=0.45, return_counts=False):
    """layers/functions/detection.py:25-70 -> (B, C, top_k, 5) float32."""
    loc, conf, priors = _f32(loc), _f32(conf), _f32(priors)
    B, P = loc.shape[0], priors.shape[0]
    Cn = num_classes
    assert conf.size == B * P * Cn and loc.size == B * P * 4
    scale = _f32(scale)
    out = np.empty((B, Cn, top_k, 5), np.float32)
    counts = np.zeros((B, Cn), np.int32)
    arm = _f32(arm_loc) if arm_loc is not None else None
    lib().orc_detect(loc.ctypes.data_as(C.c_void_p), conf.ctypes.data_as(C.c_void_p),
                     priors.ctypes.data_as(C.c_void_p),
                     arm.ctypes.data_as(C.c_void_p) if arm is not None else None,
                     scale.ctypes.data_as(C.c_void_p), B, P, Cn, top_k,
                     C.c_float(conf_thresh), C.c_double(float(nms_thresh)),
                     out.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p))
    return (out, counts) if return_counts else out


def l2norm(x, weight, eps=1e-10):
    """layers/modules/l2norm.py:17-21, NCHW."""
    x, weight = _f32(x), _f32(weight)
    N, Cc = x.shape[:2]
    out = np.empty_like(x)
    lib().orc_l2norm(x.ctypes.data_as(C.c_void_p), weight.ctypes.data_as(C.c_void_p), N, Cc,
                     int(np.prod(x.shape[2:])), C.c_float(eps), out.ctypes.data_as(C.c_void_p))
    return out


def softmax_rows(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_softmax_rows(x.ctypes.data_as(C.c_void_p), x.shape[0], x.shape[1],
                           out.ctypes.data_as(C.c_void_p))
    return out


def base_transform_u8(image, size, mean_bgr, to_rgb=False):
    """data/__init__.py:7-12 + data/voc0712.py:467-468 for uint8 BGR frames (B,H,W,3) -> (B,3,S,S) fp32.
    cv2.resize(INTER_LINEAR, 8-bit) restated from OpenCV imgproc/resize.cpp (11-bit fixed-point coefficients;
    vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).  PARITY UNPINNED BY THE REFERENCE: cv2 is not
    installed in the build image, so no output of cv2 itself exists to compare with.  What pins it instead are
    known-answer cases worked by hand from OpenCV's algorithm (tests/test_oracle_pin.py: identity, exact 2:1 = rounded
    block mean, exact 1:2 = (512, 1536)/2048 weights with border clamp, a 3x3 -> 2x2 case through both fixed-point passes), and -- from
    outside this repo -- the sampling geometry: within ONE grey level of torch's independent floating-point bilinear
    (F.interpolate, align_corners=False) on random frames of six shape pairs."""
    img = np.asarray(image, np.uint8)
    if img.ndim == 3:
        img = img[None]
    B, H0, W0, _ = img.shape

    def coef(n_dst, n_src):
        d = np.arange(n_dst)
        f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0, n_src - 1
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, n_src - 1), c0, c1
    x0, x1, a0, a1 = coef(size, W0)
    y0, y1, b0, b1 = coef(size, H0)
    src = img.astype(np.int64)
    h = src[:, :, x0, :] * a0[None, None, :, None] + src[:, :, x1, :] * a1[None, None, :, None]     # (B,H0,S,3)
    v = (((b0[None, :, None, None] * (h[:, y0] >> 4)) >> 16) + ((b1[None, :, None, None] * (h[:, y1] >> 4)) >> 16) + 2) >> 2
    v = np.clip(v, 0, 255).astype(np.float32) - np.asarray(mean_bgr, np.float32)[None, None, None, :]
    if to_rgb:
        v = v[..., ::-1]
    return np.ascontiguousarray(v.transpose(0, 3, 1, 2)).astype(np.float32)
