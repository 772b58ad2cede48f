"""Times Detect.forward alone (hipEvents) on the bench workload's own net outputs and on the D6/D8 regimes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth

dev = torch.device("cuda:0")
B = 32
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
det = Detect(21, 0, 200, 0.01, 0.45)
cases = {}
net = build_net("test", 320, 21, 1024, 1, True, True)
net.set_compute_dtype("bf16")
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.eval().to(dev)
arm, _, odm, conf = net(torch.from_numpy(synth.synth_frames(B, 320, 100)).to(dev))
cases["net-output"] = (odm, conf, arm)
for tag, bias in (("D6", 6.0), ("D8", 8.0)):
    l, a, c = synth.synth_detect_inputs(B, 6375, 21, bias, 1)
    cases[tag] = (torch.from_numpy(l).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(a).to(dev))
for tag, (l, c, a) in cases.items():
    for _ in range(3):
        out = det.forward(l, c, pri, arm_loc_data=a, scale=[500., 375., 500., 375.])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = det.forward(l, c, pri, arm_loc_data=a, scale=[500., 375., 500., 375.])
    e1.record()
    torch.cuda.synchronize()
    cand = int((c.view(B, -1, 21)[:, :, 1:] > 0.01).sum())
    print("%-10s %.3f ms/call  candidates/(img,class) %.0f  detections %d" % (tag, e0.elapsed_time(e1) / 10, cand / (B * 20), int((out[..., 0] > 0).sum())))
