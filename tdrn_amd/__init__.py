"""tdrn_amd -- MI355X-native inference path of TDRN's dual-refinement detector.

The sub-packages mirror the reference's top-level modules (`model`, `layers`, `data`, `utils`)
so its drivers can switch with `sys.path.insert(0, <this directory>)`; all
arithmetic runs in libtdrn_hip.so (hand-written HIP for gfx950) behind the C ABI of
include/tdrn_hip.h.  There is no CPU fallback: a missing library raises at first use.
"""
__version__ = "0.1"
