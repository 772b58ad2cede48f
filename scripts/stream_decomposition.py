"""where the streamed mode's 10-15 % goes: step graph without copies / with D2H / with eager H2D on its own stream"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from tdrn_amd.data import mb_cfg, base_transform
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth
from tdrn_amd.engine import GraphedCall
dev = torch.device("cuda", 0)
B = 32
net = build_net("test", 320, 21, 1024, 1, True, True)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval(); net.set_compute_dtype("bf16")
eng = net.engine(dev)
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
det = Detect(21, 0, 200, 0.01, 0.45)
rng = np.random.RandomState(7)
x = torch.from_numpy(synth.synth_frames(B, 320, seed=100)).to(dev)
scale = [500.0, 375.0, 500.0, 375.0]
def step_res(xin):
    r = eng.forward(xin); return det.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
def step_u8(u8):
    return step_res(base_transform(u8, 320, (104.0, 117.0, 123.0)))
def t(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
g0 = GraphedCall(step_res, x)
base = t(lambda: g0(g0.inputs[0]))
print("V0 resident graph               %.3f ms" % base)
NS = 4
host_in = [torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3), dtype=np.uint8)).pin_memory() for _ in range(NS)]
dev_in = [h.to(dev) for h in host_in]
gs = [GraphedCall(step_u8, d) for d in dev_in]
k = [0]
def v1():
    s = k[0] % NS; k[0] += 1
    gs[s](gs[s].inputs[0])
r = t(v1); print("V1 + preprocess, no copies      %.3f ms  (%.3f of resident)" % (r, base / r))
host_out = [torch.empty(tuple(gs[0].outputs.shape), dtype=gs[0].outputs.dtype).pin_memory() for _ in range(NS)]
def v2():
    s = k[0] % NS; k[0] += 1
    gs[s](gs[s].inputs[0])
    host_out[s].copy_(gs[s].outputs, non_blocking=True)
r = t(v2); print("V2 + D2H after the graph        %.3f ms  (%.3f)" % (r, base / r))
cs = torch.cuda.Stream(dev)
def v3():
    s = k[0] % NS; k[0] += 1
    with torch.cuda.stream(cs):
        gs[(s + 2) % NS].inputs[0].copy_(host_in[(s + 2) % NS], non_blocking=True)     # no ordering at all (timing probe)
    gs[s](gs[s].inputs[0])
r = t(v3); print("V3 V1 + unordered eager H2D     %.3f ms  (%.3f)" % (r, base / r))
def v4():
    s = k[0] % NS; k[0] += 1
    with torch.cuda.stream(cs):
        gs[(s + 2) % NS].inputs[0].copy_(host_in[(s + 2) % NS], non_blocking=True)
    gs[s](gs[s].inputs[0])
    host_out[s].copy_(gs[s].outputs, non_blocking=True)
r = t(v4); print("V4 V3 + D2H                     %.3f ms  (%.3f)" % (r, base / r))
ds = torch.cuda.Stream(dev)
ev_step = [torch.cuda.Event() for _ in range(NS)]
ev_in = [torch.cuda.Event() for _ in range(NS)]
def v5():   # ordered: H2D two ahead waits for the step that last read that slot; the step waits for its own H2D; D2H on its own stream
    s = k[0] % NS; k[0] += 1
    nxt = (s + 2) % NS
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(cs):
        cs.wait_event(ev_step[nxt])
        gs[nxt].inputs[0].copy_(host_in[nxt], non_blocking=True)
        ev_in[nxt].record(cs)
    cur.wait_event(ev_in[s])
    gs[s](gs[s].inputs[0])
    ev_step[s].record(cur)
    with torch.cuda.stream(ds):
        ds.wait_event(ev_step[s])
        host_out[s].copy_(gs[s].outputs, non_blocking=True)
for e in ev_step + ev_in: e.record(torch.cuda.current_stream(dev))
r = t(v5); print("V5 ordered: H2D 2 ahead, D2H on own stream  %.3f ms  (%.3f)" % (r, base / r))
def v6():   # copy only
    s = k[0] % NS; k[0] += 1
    with torch.cuda.stream(cs):
        gs[s].inputs[0].copy_(host_in[s], non_blocking=True)
r = t(v6); print("V6 H2D alone                    %.3f ms" % r)
print("V0 again                        %.3f ms" % t(lambda: g0(g0.inputs[0])))
from tdrn_amd.stream import FrameStream
for slots in (3, 4, 3, 4):
    fs = FrameStream(eng, det, pri, B, slots=slots); print(fs.calibration["picked"], sorted(fs.calibration["ms_per_step"].values())[::5])
    for s in range(slots):
        fs.pinned_in(s).copy_(host_in[s % NS])
    fs.prime()
    r = t(fs.run); print("FrameStream slots=%d             %.3f ms  (%.3f)" % (slots, r, base / r))
    fs.drain()
print("V0 again                        %.3f ms" % t(lambda: g0(g0.inputs[0])))
r = t(v5); print("V5 again  %.3f ms  (%.3f)" % (r, base / r))
