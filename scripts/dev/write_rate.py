"""how fast can this box WRITE: fill / copy of Y-sized buffers (370 MB), hipEvent-timed"""
import torch
dev = torch.device("cuda:0")
n = 370 * 1000 * 1000 // 4
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
us = timed(lambda: a.fill_(1.0)); print("fill 370 MB: %.1f us = %.2f TB/s written" % (us, 0.37e9 / us / 1e6))
us = timed(lambda: b.copy_(a)); print("copy 370 MB: %.1f us = %.2f TB/s written (+ the same read)" % (us, 0.37e9 / us / 1e6))
us = timed(lambda: torch.add(a, 1.0, out=b)); print("add  370 MB: %.1f us = %.2f TB/s written (+ the same read)" % (us, 0.37e9 / us / 1e6))
