"""ctypes binding of libtdrn_hip.so (include/tdrn_hip.h).  Fails loudly: no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDRN_LIB_PATH") or os.path.join(_HERE, "lib", "libtdrn_hip.so")   # override: diagnostics builds only

F32, BF16, F16 = 0, 1, 2
DTYPES = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "fp16": F16,
          "f16": F16, "float16": F16, "half": F16}
DRN_VGGBN, DRN_MOBILENET, SSD4SCALE_MOBILE, REFINEDET_VGG, SSD4SCALE_VGG = range(5)

E_VALUE = -7


class TdrnError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("%s (tdrn error %d)%s" % (error_string(code), code,
                                                    (": " + what) if what else ""))


class NetConfig(C.Structure):
    _fields_ = [("model", C.c_int), ("size", C.c_int), ("num_classes", C.c_int),
                ("c7_channel", C.c_int), ("def_groups", C.c_int), ("bn", C.c_int),
                ("multihead", C.c_int), ("deform", C.c_int), ("test_phase", C.c_int),
                ("dtype", C.c_int), ("use_refine", C.c_int), ("plan_flags", C.c_int), ("reserved", C.c_int * 4)]


PLAN_NO_FUSE_FIRST, PLAN_NO_LATE_SIDE, PLAN_ONE_STREAM, PLAN_NO_DEFORM_TS, PLAN_CHAIN = 1, 2, 4, 8, 16      # tdrn_hip.h TDRN_PLAN_*
PLAN_NO_CONV_PP, PLAN_NO_PP_SK, PLAN_NO_CONV_PATCH, PLAN_FAULT_HANDOFF, PLAN_DWPW, PLAN_NO_PW1X1 = 32, 64, 128, 256, 512, 1024
PLAN_NO_DW_SLIDE, PLAN_DW_SLIDE_ALL, PLAN_NO_CONV_WS, PLAN_NO_YGEMM_V2, PLAN_NO_HEAD3X3, PLAN_TS_ONE_RANGE = 2048, 4096, 8192, 16384, 32768, 65536
PLAN_NO_PATCH_TAIL = 131072
E_DEVICE = -8


class NetIO(C.Structure):
    _fields_ = [("x", C.c_void_p), ("batch", C.c_int), ("arm_loc", C.c_void_p),
                ("odm_loc", C.c_void_p), ("conf", C.c_void_p), ("offsets", C.c_void_p * 4),
                ("ref_loc", C.c_void_p * 4), ("loc_maps", C.c_void_p * 4),
                ("reserved", C.c_void_p * 4)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int), ("flops", C.c_double),
                ("bytes", C.c_double), ("ms", C.c_double)]


class U8FramesABI(C.Structure):
    """tdrn_u8_frames (tdrn_hip.h): the batch as uint8 planes + the per-plane mean (tdrn_net_io.reserved[3])"""
    _fields_ = [("planes", C.c_void_p), ("mean", C.c_float * 3)]


class OpInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_", C.c_int), ("out", C.c_int), ("res", C.c_int), ("pool", C.c_int), ("off", C.c_int),
                ("y", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("dil", C.c_int), ("relu", C.c_int),
                ("ceil_mode", C.c_int), ("splitk", C.c_int), ("groups", C.c_int), ("out_kind", C.c_int), ("level", C.c_int),
                ("n_branches", C.c_int), ("k2", C.c_int), ("pad2", C.c_int), ("off_c0", C.c_int * 2), ("y_tap_major", C.c_int),
                ("fused_first", C.c_int), ("fused_dw", C.c_int), ("w", C.c_char * 48), ("b", C.c_char * 48), ("bn", C.c_char * 48), ("w2", C.c_char * 48),
                ("b2", C.c_char * 48), ("y_groups", C.c_int)]


OP_KINDS = ("first_conv", "conv", "conv_transpose", "depthwise", "maxpool", "l2norm", "offset_conv", "deform_heads", "other")

_lib = None

_SIGS = {
    "tdrn_version": (C.c_char_p, []),
    "tdrn_error_string": (C.c_char_p, [C.c_int]),
    "tdrn_deform_conv_workspace_bytes": (C.c_size_t, [C.c_int] * 15),
    "tdrn_deform_conv_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 15 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_nms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "tdrn_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_nms_topk": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_gpu_nms_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]),
    "tdrn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "tdrn_center_size": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "tdrn_prior_box": (C.c_int, [C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tdrn_nms_topk_classes_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tdrn_nms_topk_classes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_roi_resample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tdrn_ota_similarity": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "tdrn_preprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "tdrn_preprocess_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tdrn_detect_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "tdrn_detect": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_double, C.c_void_p,
                                                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_detect_dev_scale": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_double, C.c_void_p,
                                                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_net_create": (C.c_int, [C.POINTER(NetConfig), C.POINTER(C.c_void_p)]),
    "tdrn_net_destroy": (None, [C.c_void_p]),
    "tdrn_net_param_count": (C.c_int, [C.c_void_p]),
    "tdrn_net_param_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64 * 4),
                                      C.POINTER(C.c_int)]),
    "tdrn_net_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "tdrn_net_weight_bytes": (C.c_size_t, [C.c_void_p]),
    "tdrn_net_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "tdrn_net_num_priors": (C.c_int, [C.c_void_p]),
    "tdrn_net_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tdrn_net_adopt_weights": (C.c_int, [C.c_void_p]),
    "tdrn_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(NetIO), C.c_void_p]),
    "tdrn_net_check": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint)]),
    "tdrn_net_op_count": (C.c_int, [C.c_void_p]),
    "tdrn_net_op_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(OpInfo)]),
    "tdrn_net_tensor_count": (C.c_int, [C.c_void_p]),
    "tdrn_net_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tdrn_net_read_tensor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tdrn_net_write_tensor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tdrn_net_forward_from": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(NetIO), C.c_int, C.c_void_p]),
    "tdrn_net_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "tdrn_net_kernel_stats": (C.c_int, [C.c_void_p, C.POINTER(KernelStat), C.c_int]),
    "tdrn_net_op_stats": (C.c_int, [C.c_void_p, C.POINTER(KernelStat), C.c_int]),
    "tdrn_net_op_timeline": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libtdrn_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C tdrn_amd/csrc`.  tdrn_amd has no CPU fallback." % LIB_PATH)
        # Load order matters on a GPU box: the library registers its code objects with the HIP runtime when it is
        # mapped, and if that happens before PyTorch has brought the runtime up, every later HIP call from the
        # library fails with hipErrorNoDevice (seen with build() followed by smoke() in one process).  So let
        # torch initialise the device first whenever one is visible (a no-op in the GPU-less build container).
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)           # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def error_string(code):
    try:
        return lib().tdrn_error_string(int(code)).decode()
    except Exception:
        return "tdrn error"


def check(rc, what=""):
    if rc != 0:
        if rc == E_VALUE:
            raise ValueError("nms_threshold must be non negative.")
        raise TdrnError(rc, what)


def ptr(t):
    """Device/host address of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise NotImplementedError(
            "%s must live on the GPU: tdrn_amd runs only through libtdrn_hip.so on an MI355X "
            "(no CPU path; the reference's ConvOffset2dFunction raises here too, "
            "model/networks.py:632-633)" % name)

