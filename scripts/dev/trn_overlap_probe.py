"""Which part of the static-net-beside-the-temporal-trunk pattern takes hipStreamEndCapture down?  python scripts/dev/trn_overlap_probe.py <variant>"""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tdrn_amd.utils import synth
import importlib

variant = sys.argv[1]
dev = torch.device("cuda:0")
def make(deform, seed):
    net = importlib.import_module("tdrn_amd.model.ssd4scale_vgg").build_net("test", 320, 21, c7_channel=1024, bn=False, deform=deform)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval(); net.set_compute_dtype("bf16"); net.engine(dev)
    return net
from tdrn_amd import _lib
stat, temp = make(False, 0), make(True, 1)
if "one_stream" in variant:
    stat = importlib.import_module("tdrn_amd.model.ssd4scale_vgg").build_net("test", 320, 21, c7_channel=1024, bn=False, deform=False)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in stat.state_dict().items()}, 0)
    stat.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    stat.eval(); stat.set_plan_flags(_lib.PLAN_ONE_STREAM); stat.set_compute_dtype("bf16"); stat.engine(dev)
B, F = 2, 4
clips = torch.from_numpy(synth.synth_frames(B * F, 320, seed=3)).to(dev).view(F, B, 3, 320, 320)
side, ev = torch.cuda.Stream(dev), torch.cuda.Event()

def step():
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        s_loc, _, maps = stat(clips[0], ret_loc=True)
        ev.record(side)
    if variant.startswith("stat_only"):
        main.wait_stream(side)
        return s_loc, maps[0]
    if variant.startswith("in_net_wait"):
        loc, conf = temp(clips.view(F * B, 3, 320, 320), ref_loc=maps, ref_event=ev)[:2]
    else:
        main.wait_event(ev)
        loc, conf = temp(clips.view(F * B, 3, 320, 320), ref_loc=maps)[:2]
    main.wait_stream(side)
    if variant.endswith("+rs") or variant == "rs":
        s_loc.record_stream(main)
    return loc, conf, s_loc

want = [t.clone() for t in step()]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print("captured", variant, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed; equal:", [bool(torch.equal(a, b)) for a, b in zip(out, want)], flush=True)
