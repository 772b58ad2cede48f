"""Where does the streamed mode (FrameStream) lose time against the resident replay?  hipEvents around every copy-in, step and
copy-out of a few steady-state steps, on the streams they run on, printed as one timeline (ms from the first event).
rocprofv3 cannot show this: its kernel trace serialises the dispatches.

    python scripts/dev/stream_timeline.py [NP pipelines] [steps shown]      env: CAL=0 skips the stream-pair calibration
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.stream import FrameStream, AHEAD
from tdrn_amd.utils import synth

dev = torch.device("cuda:0")
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SHOW = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 32
net = build_net("test", 320, 21, 1024, 1, True, True)
net.set_compute_dtype("bf16")
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval()
eng = net.engine(dev)
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
engines = [eng] + [eng.clone() for _ in range(NP - 1)]
NSL = 4
fs = FrameStream(engines if NP > 1 else eng, Detect(21, 0, 200, 0.01, 0.45), pri, B, slots=NSL, calibrate=os.environ.get("CAL", "1") != "0", graph=os.environ.get("MODE", "graph") != "eager", zero_copy_out=os.environ.get("ZC", "0") != "0", copy_in=os.environ.get("CI", "stream"))
rng = np.random.RandomState(7)
for sl in range(NSL):
    fs.pinned_in(sl).copy_(torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3), dtype=np.uint8)))
print("calibration:", fs.calibration and fs.calibration["picked"])


def E():
    return torch.cuda.Event(enable_timing=True)


rec = []          # (label, step, start event, end event)
orig_copy_in = fs._copy_in


def run_instrumented(k_show):
    s = fs._k % fs.slots
    slot_in = (fs._k + AHEAD) % fs.slots
    # copy-in
    a, b = E(), E()
    with torch.cuda.stream(fs._in_stream):
        fs._in_stream.wait_event(fs.ev_step[slot_in])
        fs._in_stream.wait_event(fs.ev_out[slot_in])
        a.record(fs._in_stream)
        fs.dev_in[slot_in].copy_(fs.host_in[slot_in], non_blocking=True)
        fs.ev_in[slot_in].record(fs._in_stream)
        b.record(fs._in_stream)
    rec.append(("H2D for step %d" % (fs._k + AHEAD), fs._k, a, b))
    p = s % fs.pipelines
    cur = torch.cuda.current_stream(dev) if p == 0 else fs._extra_streams[p - 1]
    a, b = E(), E()
    with torch.cuda.stream(cur):
        cur.wait_event(fs.ev_in[s])
        a.record(cur)
        if fs.graph:
            fs.graphs[s].replay()
        elif fs.zero_copy_out:
            fs._steps[p](fs.dev_in[s], out=fs.host_out[s])
        else:
            fs.dev_out[s].copy_(fs._steps[p](fs.dev_in[s]))
        fs.ev_step[s].record(cur)
        if fs.zero_copy_out:
            fs.ev_out[s].record(cur)
        b.record(cur)
    rec.append(("step %d (pipeline %d)" % (fs._k, p), fs._k, a, b))
    a, b = E(), E()
    if fs.zero_copy_out:
        fs._step_of[s] = fs._k
        fs._k += 1
        return
    with torch.cuda.stream(fs._out_stream):
        fs._out_stream.wait_event(fs.ev_step[s])
        a.record(fs._out_stream)
        fs.host_out[s].copy_(fs.dev_out[s], non_blocking=True)
        fs.ev_out[s].record(fs._out_stream)
        b.record(fs._out_stream)
    rec.append(("D2H of step %d" % fs._k, fs._k, a, b))
    fs._step_of[s] = fs._k
    fs._k += 1


import time
fs.prime()
for _ in range(12):
    fs.run()
fs.drain()
# plain rate
t0 = time.perf_counter()
for _ in range(60):
    fs.run()
fs.drain()
dt = (time.perf_counter() - t0) / 60
print("streamed (%s, copy-in %s, zero-copy out %s, NP %d): %.3f ms per step = %.0f frames/s" % ("graph" if fs.graph else "eager", fs.copy_in, fs.zero_copy_out, NP, dt * 1e3, B / dt))
if fs.copy_in == "own" or SHOW <= 0:
    sys.exit(0)                                          # (the instrumented replica below knows the copy-in stream only)
base = E()
fs.prime()
for _ in range(8):
    fs.run()
base.record(torch.cuda.current_stream(dev))
for _ in range(SHOW):
    run_instrumented(0)
fs.drain()
rows = [(base.elapsed_time(a), base.elapsed_time(b), lab) for lab, k, a, b in rec]
rows.sort()
for t0_, t1_, lab in rows:
    print("%8.3f -> %8.3f  (%6.3f ms)  %s" % (t0_, t1_, t1_ - t0_, lab))
