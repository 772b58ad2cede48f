"""Yardstick (not a baseline of the contract, not product code): what the reference's OWN forward -- PyTorch ops, here torch 2.10 + MIOpen on the
same MI355X -- takes for the parts it can run on this GPU: the VGG16-BN trunk + extras + TCB / ARM heads of dualrefinedet_vggbn (everything but
the deformable ODM heads, whose op exists only as a CUDA extension in the reference), batch 32, 16-bit, channels-last, eager and under
torch.compile-free CUDA graphs.  Prints ms per batch next to this repo's forward of the whole net.

    python scripts/torch_gpu_yardstick.py        (on the GPU box)
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import net_ref                                   # noqa: E402  (the restated forward: plain torch ops)
from tdrn_amd.model.dualrefinedet_vggbn import build_net     # noqa: E402
from tdrn_amd.utils import synth                             # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = 32
    net = build_net("test", 320, 21, 1024, 1, True, True)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    x32 = torch.from_numpy(synth.synth_frames(B, 320, seed=100)).to(dev)
    for dt in (torch.bfloat16, torch.float16):
        sdt = {k: (torch.from_numpy(v).to(dev).to(dt) if v.dtype.kind == "f" else torch.from_numpy(v).to(dev)) for k, v in sd.items()}
        x = x32.to(dt).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            t_trunk = timed(lambda: net_ref.vgg_trunk(sdt, x, True))
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    net_ref.vgg_trunk(sdt, x, True)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                out = net_ref.vgg_trunk(sdt, x, True)
            t_graph = timed(g.replay)
        print("%s: the reference's VGG16-BN trunk (conv1_1 .. fc7) as plain torch ops on this GPU, batch %d: %.2f ms eager, %.2f ms as one CUDA graph "
              "(%.0f frames/s for the trunk alone)" % (str(dt).split(".")[-1], B, t_trunk, t_graph, B / t_graph * 1e3))
    net.set_compute_dtype("bf16")
    eng = net.engine(dev)
    t_ours = timed(lambda: eng.forward(x32))
    print("this repo, bf16: the WHOLE forward (trunk + extras + TCB + ARM + deformable ODM heads + softmax), batch %d: %.2f ms (one step at a time, eager)" % (B, t_ours))


if __name__ == "__main__":
    main()
