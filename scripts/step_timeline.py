"""Timeline of one eager forward in the production schedule (events recorded by the library on each lane)."""
import sys, torch
sys.path.insert(0, ".")
from tdrn_amd.model import dualrefinedet_vggbn as m
from tdrn_amd.utils import synth
dev = torch.device("cuda:0")
net = m.build_net("test", 320, 21, 1024, 1, True, True)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 0).items()})
net.eval().to(dev).bfloat16()
x = torch.from_numpy(synth.synth_frames(32, 320, seed=1)).to(dev)
eng = net.engine(dev) if hasattr(net, "engine") else net._engine
for _ in range(3):
    net(x)
eng.set_profile(2)
for _ in range(2):
    net(x)
torch.cuda.synchronize()
tl = eng.op_timeline()
eng.set_profile(0)
for o in sorted(tl, key=lambda o: o["start"]):
    print("%8.1f %8.1f %7.1f  lane %d  %s" % (o["start"] * 1e3, o["end"] * 1e3, (o["end"] - o["start"]) * 1e3, o["lane"], o["name"]))
