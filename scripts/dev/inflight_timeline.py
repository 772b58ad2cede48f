"""hipEvents before / after every replay of the two-steps-in-flight schedule (engine.InFlight, resident batches): does a pipeline's next
step start when its previous one ends?  (rocprofv3's kernel trace serialises dispatches and cannot show it.)

    python scripts/dev/inflight_timeline.py [NP] [NB resident batches] [steps shown]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tdrn_amd.data import mb_cfg
from tdrn_amd.engine import InFlight
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth

dev = torch.device("cuda:0")
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SHOW = int(sys.argv[3]) if len(sys.argv) > 3 else 12
B = 32
net = build_net("test", 320, 21, 1024, 1, True, True)
net.set_compute_dtype("bf16")
if int(os.environ.get("FLAGS", "0")):
    net.set_plan_flags(int(os.environ["FLAGS"]))
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval()
eng = net.engine(dev)
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
xb = [torch.from_numpy(synth.synth_frames(B, 320, seed=100 + i)).to(dev) for i in range(NB)]


def make_step(e):
    d = Detect(21, 0, 200, 0.01, 0.45)

    def one(xin):
        r = e.forward(xin)
        return d.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=[500.0, 375.0, 500.0, 375.0])
    return one


def dummy_streams(n):
    """n HIP streams nobody uses: shifts the round-robin assignment of the NEXT streams to the hardware queues (experiment)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    for _ in range(n):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
        KEEP.append(h)


KEEP = []
dummy_streams(int(os.environ.get("SHIFT_A", "0")))
eng.forward(xb[0])                                       # engine 0's side lanes exist from here on
torch.cuda.synchronize()
dummy_streams(int(os.environ.get("SHIFT_B", "0")))
fl = InFlight(make_step, eng, xb, n=NP, graph=os.environ.get("MODE", "graph") != "eager")
for k in range(3 * NB):
    fl.launch(k)
fl.sync()
if os.environ.get("PICK", "0") != "0":
    cal = fl.pick_streams()
    if cal:
        ms = cal["ms_per_step"]
        print("stream pairs (ms per step):", " ".join("%s:%.3f" % kv for kv in sorted(ms.items(), key=lambda kv: kv[1])), "-> picked", cal["picked"])
t0 = time.perf_counter()
for k in range(100):
    fl.launch(k)
fl.sync()
dt = (time.perf_counter() - t0) / 100
print("shift %s/%s NP %d NB %d %s queues %s flags %s grid %s: %.3f ms per step = %.0f frames/s" % (os.environ.get("SHIFT_A", "0"), os.environ.get("SHIFT_B", "0"), NP, NB, os.environ.get("MODE", "graph"), os.environ.get("GPU_MAX_HW_QUEUES", "4"), os.environ.get("FLAGS", "0"), os.environ.get("TDRN_MAIN_GRID", "-"), dt * 1e3, B / dt))
if os.environ.get("OPS", "0") != "0" and not fl.graph:
    # per-launch hipEvents on the production lanes of BOTH pipelines while they overlap: where does a step wait?
    eng.set_profile(1)
    eng.forward(xb[0]); eng.forward(xb[0])
    torch.cuda.synchronize()
    alone = {o["name"]: o["ms"] for o in eng.op_stats()}
    for e in fl.engines:
        e.set_profile(2)
    for k in range(3 * NB):
        fl.launch(k)
    fl.sync()
    tl = fl.engines[0].op_timeline()
    for e in fl.engines:
        e.set_profile(0)
    prev_end = {}
    print("%-42s lane %9s %9s %8s %8s %8s" % ("launch (pipeline 0, other pipeline running)", "start", "end", "us", "alone", "gap"))
    for o in tl:
        gap = (o["start"] - prev_end.get(o["lane"], o["start"])) * 1e3
        print("%-42s %4d %9.3f %9.3f %8.1f %8.1f %8.1f" % (o["name"], o["lane"], o["start"], o["end"], o["ms"] * 1e3, alone.get(o["name"], 0) * 1e3, gap))
        prev_end[o["lane"]] = o["end"]
if SHOW <= 0:
    sys.exit(0)
E = lambda: torch.cuda.Event(enable_timing=True)
for k in range(2 * NB):
    fl.launch(k)
base = E()
base.record(torch.cuda.current_stream(dev))
rec = []
for k in range(SHOW):
    j = k % NB
    st = fl.stream_of(j)
    a, b = E(), E()
    a.record(st)
    fl.launch(k)
    b.record(st)
    rec.append((k, j % NP, a, b))
fl.sync()
for k, p, a, b in rec:
    print("step %2d pipeline %d batch %d: %8.3f -> %8.3f (%.3f ms)" % (k, p, k % NB, base.elapsed_time(a), base.elapsed_time(b), a.elapsed_time(b)))
