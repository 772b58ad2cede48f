"""Forward-only throughput of every model family the engine plans (information beside bench.py, which
measures BASELINE.json's metric only).  python scripts/bench_models.py [--dtype bf16]"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tdrn_amd.utils import synth

CASES = [  # module, build_net args after phase (positional), keyword args, batch (BASELINE.json configs 2-5)
    ("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True), {}, 32),
    ("dualrefinedet_vggbn", (512, 21, 1024, 1, True, True), {}, 16),
    ("dualrefinedet_mobilenet", (320, 21), dict(def_groups=1, multihead=True), 64),
    ("refinedet_vgg", (320, 21), dict(use_refine=True, bn=False), 32),
    ("ssd4scale_vgg", (320, 21), dict(bn=False, deform=False), 32),
    ("ssd4scale_mobile", (320, 21), dict(deform=False), 64),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--per-op", default="", help="module name: print that model's per-launch timings")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for mod, a, kw, B in CASES:
        net = importlib.import_module("tdrn_amd.model." + mod).build_net("test", *a, **kw)
        net.set_compute_dtype(args.dtype)
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net = net.eval().to(dev)
        x = torch.from_numpy(synth.synth_frames(B, a[0], seed=1)).to(dev)
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print("%-26s %4d px  batch %3d  %s  %8.1f frames/s  %7.3f ms/batch" % (mod, a[0], B, args.dtype, B / dt, dt * 1e3))
        if args.per_op == mod:
            eng = net._engine
            eng.set_profile(True)
            net(x); net(x)
            torch.cuda.synchronize()
            for o in eng.op_stats():
                tf = o["flops"] / (o["ms"] * 1e-3) / 1e12 if o["ms"] > 0 else 0.0
                print("    %-44s %8.1f us %8.1f GF %7.1f TF/s %7.2f GB %7.0f GB/s" % (o["name"], o["ms"] * 1e3, o["flops"] / 1e9, tf, o["bytes"] / 1e9, o["bytes"] / 1e9 / (o["ms"] * 1e-3) if o["ms"] > 0 else 0))
            eng.set_profile(False)


if __name__ == "__main__":
    main()
