"""Soak of the N-steps-in-flight schedule (round 5): NP pipelines (own engine + workspace + hipGraph + HIP stream each, one packed weight
blob shared) replayed round-robin for STEPS replays; every pipeline's detections are compared with the ones it produced alone
(bit for bit) every CHECK replays, and tdrn_net_check is read at the end (a chained-split hand-off that timed out fails the run).

    python scripts/dev/in_flight_soak.py NP STEPS [CHECK]     env: MODE=graph|eager, FLAGS=<plan flags of every engine>
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tdrn_amd.data import mb_cfg
from tdrn_amd.engine import GraphedCall, NetEngine
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth

dev = torch.device("cuda:0")
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
CHECK = int(sys.argv[3]) if len(sys.argv) > 3 else 500
B = int(os.environ.get("BATCH", "32"))
FLAGS = int(os.environ.get("FLAGS", "0"))
EAGER = os.environ.get("MODE", "graph") == "eager"

net = build_net("test", 320, 21, 1024, 1, True, True)
net.set_compute_dtype(os.environ.get("DTYPE", "bf16"))
if FLAGS:
    net.set_plan_flags(FLAGS)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net.eval()
eng0 = net.engine(dev)
pri = PriorBox(mb_cfg["VOC_320"]).forward().to(dev)
engines, streams, graphs, fns = [eng0], [torch.cuda.Stream(dev) for _ in range(NP)], [], []
for i in range(1, NP):
    e = NetEngine(**dict(net._engine_args, dtype=net.compute_dtype))
    e.share_weights(eng0)
    engines.append(e)
for i in range(NP):
    det = Detect(21, 0, 200, 0.01, 0.45)
    x = torch.from_numpy(synth.synth_frames(B, 320, seed=100 + i)).to(dev)

    def one(xin, e=engines[i], d=det):
        r = e.forward(xin)
        return d.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=[500.0, 375.0, 500.0, 375.0])
    fns.append(one)
    with torch.cuda.stream(streams[i]):
        g = GraphedCall(one, x)
    torch.cuda.synchronize()
    graphs.append(g)


def launch(i):
    with torch.cuda.stream(streams[i]):
        if EAGER:
            return fns[i](graphs[i].inputs[0])
        graphs[i].graph.replay()
        return graphs[i].outputs


# reference: every pipeline alone
refs = []
for i in range(NP):
    out = launch(i)
    torch.cuda.synchronize()
    refs.append(out.clone())

bad = 0
t0 = time.perf_counter()
last = [None] * NP
for k in range(STEPS):
    i = k % NP
    last[i] = launch(i)
    if (k + 1) % CHECK == 0:
        torch.cuda.synchronize()
        for j in range(NP):
            if last[j] is not None and not torch.equal(last[j], refs[j]):
                bad += 1
                print("replay %d: pipeline %d differs from its stand-alone result (max |d| %.3e)" % (k, j, float((last[j] - refs[j]).abs().max())), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for e in engines:
    e.check()
print("soak: %d pipelines, %d replays (%s), %.1f frames/s incl. the checks, mismatching checks: %d, tdrn_net_check clean" % (
    NP, STEPS, "eager" if EAGER else "graph", B * STEPS / dt, bad), flush=True)
sys.exit(1 if bad else 0)
