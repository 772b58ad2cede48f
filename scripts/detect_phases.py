"""Phase breakdown of detect_select_nms_kernel (needs a -DTDRN_DETECT_TIMING build of the library:
make -C tdrn_amd/csrc BUILD=../../_build_adt/obj OUT=../../_build_adt/libtdrn_dt.so EXTRA=-DTDRN_DETECT_TIMING
and TDRN_LIB_PATH=_build_adt/libtdrn_dt.so).  Stamps are 100-MHz wall-clock ticks per (image, class) workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth

dev = torch.device("cuda:0")
B, C = 32, 21
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
det = Detect(C, 0, 200, 0.01, 0.45)
net = build_net("test", 320, C, 1024, 1, True, True)
net.set_compute_dtype("bf16")
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.eval().to(dev)
arm, _, odm, conf = net(torch.from_numpy(synth.synth_frames(B, 320, 100)).to(dev))
for _ in range(3):
    out = det.forward(odm, conf, pri, arm_loc_data=arm, scale=[500., 375., 500., 375.])
torch.cuda.synchronize()
ws = det._ws.cpu().numpy()
st = ws[-B * C * 64:].view(np.int64).reshape(B * C, 8)[:, :7].astype(np.float64) * 0.01     # us
fg = np.array([s for s in range(B * C) if s % C != 0])
st = st[fg]
t0 = st[:, 0].min()
names = ["scan", "select", "compact", "sort", "nms", "pack"]
d = np.diff(st, axis=1)
print("workgroup start spread: %.1f us; kernel span %.1f us" % (st[:, 0].max() - t0, st[:, 6].max() - t0))
for i, n in enumerate(names):
    print("%-8s mean %7.1f us   max %7.1f us" % (n, d[:, i].mean(), d[:, i].max()))
print("total    mean %7.1f us   max %7.1f us" % ((st[:, 6] - st[:, 0]).mean(), (st[:, 6] - st[:, 0]).max()))
