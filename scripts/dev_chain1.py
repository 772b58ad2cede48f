"""one forward with the chain launch, one without; compare; print the chain's counters (workspace tail)"""
import sys, os, time, torch
sys.path.insert(0, ".")
from tdrn_amd import _lib
from tdrn_amd.model import dualrefinedet_vggbn as m
from tdrn_amd.utils import synth
dev = torch.device("cuda:0")
def build(flags):
    net = m.build_net("test", 320, 21, 1024, 1, True, True)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 0).items()})
    net.eval().to(dev)
    net.set_plan_flags(flags)
    net.set_compute_dtype(sys.argv[2] if len(sys.argv) > 2 else "bf16")
    return net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.from_numpy(synth.synth_frames(B, 320, seed=1)).to(dev)
a, b = build(0), build(_lib.PLAN_NO_CHAIN)
print("built", flush=True)
rb = b(x); torch.cuda.synchronize(); print("plain done", flush=True)
t = time.time(); ra = a(x); torch.cuda.synchronize(); print("chain done %.3f s" % (time.time() - t), flush=True)
eng = a._engine
ws = eng._ws if hasattr(eng, "_ws") else None
for k in dir(eng):
    v = getattr(eng, k, None)
    if torch.is_tensor(v) and v.dtype == torch.uint8 and v.numel() > 1 << 20:
        n = eng.lib.tdrn_net_workspace_bytes(eng.handle, B)
        tail = v[:n][-(n - 0):]
        print("workspace attr", k, v.numel(), n)
def flat(r):
    out = []
    for u in r:
        out += list(u) if isinstance(u, (list, tuple)) else [u]
    return out
for i, (u, v) in enumerate(zip(flat(ra), flat(rb))):
    print(i, tuple(u.shape), "equal" if torch.equal(u, v) else "DIFF max %.3g" % (u.float() - v.float()).abs().max().item())
for _ in range(5):
    t = time.time(); ra = a(x); torch.cuda.synchronize(); print("chain again %.4f s" % (time.time() - t), flush=True)
for i, (u, v) in enumerate(zip(flat(ra), flat(rb))):
    print(i, "equal" if torch.equal(u, v) else "DIFF max %.3g" % (u.float() - v.float()).abs().max().item())
