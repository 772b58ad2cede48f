"""Config 4 probe: one forward + Detect of 64 frames against TWO concurrent ones of 32 (two engines, shared weights, two streams, joined per step).
python scripts/dev/split_batch_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tdrn_amd.data import mb_cfg
from tdrn_amd.engine import GraphedCall, NetEngine
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_mobilenet import build_net
from tdrn_amd.utils import synth

dev = torch.device("cuda:0")
B, K = 64, 40
net = build_net("test", 320, 21, def_groups=1, multihead=True)
net.set_compute_dtype("bf16")
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval()
eng0 = net.engine(dev)
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
scale = [500.0, 375.0, 500.0, 375.0]
x = torch.from_numpy(synth.synth_frames(B, 320, seed=100)).to(dev)

class _Eager(object):
    def __init__(self, fn, inp):
        self.fn, self.inp = fn, inp
        self.outputs = fn(inp)
        self.graph = self
    def replay(self):
        self.outputs = self.fn(self.inp)
    def __call__(self, inp):
        self.replay()

def run_graph(fn, inp, steps=K):
    g = _Eager(fn, inp) if os.environ.get("TDRN_PROBE_EAGER", "1") == "1" else GraphedCall(fn, inp)
    for _ in range(5):
        g(inp)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            g.graph.replay()
        torch.cuda.synchronize()
        best = max(best, B * steps / (time.perf_counter() - t0))
    return best, g

det0 = Detect(21, 0, 200, 0.01, 0.45)
def whole(xin):
    r = eng0.forward(xin)
    return det0.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)

for NS in (2, 4):
    engines = [eng0]
    for i in range(1, NS):
        e = NetEngine(**dict(net._engine_args, dtype="bf16", plan_flags=int(os.environ.get("TDRN_PROBE_FLAGS", "0"))))
        e.share_weights(eng0)
        engines.append(e)
    dets = [Detect(21, 0, 200, 0.01, 0.45) for _ in range(NS)]
    sides = [torch.cuda.Stream(dev) for _ in range(NS - 1)]
    def split(xin, NS=NS, engines=engines, dets=dets, sides=sides):
        main = torch.cuda.current_stream(dev)
        parts = torch.chunk(xin, NS)
        outs = [None] * NS
        for i in range(1, NS):
            sides[i - 1].wait_stream(main)
            with torch.cuda.stream(sides[i - 1]):
                r = engines[i].forward(parts[i])
                outs[i] = dets[i].forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
        r = engines[0].forward(parts[0])
        outs[0] = dets[0].forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
        for s in sides:
            main.wait_stream(s)
        return torch.cat(outs, 0)
    a, ga = run_graph(whole, x)
    b, gb = run_graph(split, x)
    a2, _ = run_graph(whole, x)
    print("NS=%d: one forward of %d: %.0f / %.0f frames/s; %d concurrent forwards of %d: %.0f frames/s; detections equal: %s" % (
        NS, B, a, a2, NS, B // NS, b, bool(torch.equal(ga.outputs, gb.outputs))), flush=True)
