"""Round 6, review item 6, the SPEED half as far as it can be measured without writing the fused kernel: Winograd F(2x2, 3x3) of conv3_2 / conv4_2
(batch 32) in its UNFUSED form on the GPU -- input transform (a torch elementwise pass), the 16 per-position GEMMs as one batched GEMM on the
vendor library (torch.bmm = hipBLASLt: the best a GEMM can do here), output transform -- each phase timed with events, next to torch's own
direct convolution of the same layer.  A yardstick for the cost model in profiles/r06_experiments.md, not product code (the product links no
library): the unfused form must move V (4x the input) and M (4x the output) through HBM, which is what the fused kernel would have to avoid.

    python scripts/winograd_feasibility_gpu.py            (on the GPU box)
"""
import sys
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32, device=dev)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3          # us


def main():
    torch.manual_seed(0)
    for name, C, K, H, dt in (("conv3_2", 256, 256, 80, torch.bfloat16), ("conv4_2", 512, 512, 40, torch.bfloat16),
                              ("conv3_2", 256, 256, 80, torch.float16), ("conv4_2", 512, 512, 40, torch.float16)):
        B = 32
        x = torch.relu(torch.randn(B, C, H, H, device=dev)).to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5)
        w16 = w.to(dt).contiguous(memory_format=torch.channels_last)
        U = torch.einsum("ai,kcij,bj->abkc", G, w, G).reshape(16, K, C).to(dt).contiguous()          # packed once, like the weights
        th = H // 2
        T = B * th * th

        def in_tf():
            xp = F.pad(x, (1, 1, 1, 1))
            d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # (B, C, th, tw, 4, 4)
            V = torch.einsum("ai,bctxij,dj->adbtxc", BT.to(dt), d, BT.to(dt))      # (4, 4, B, th, tw, C)
            return V.reshape(16, T, C)
        V = in_tf()

        def gemm():
            return torch.bmm(V, U.transpose(1, 2))                        # (16, T, K)
        M = gemm()

        def out_tf():
            Y = torch.einsum("pa,abtk,qb->tpqk", AT.to(dt), M.reshape(4, 4, T, K), AT.to(dt))
            return Y
        t_in, t_g, t_out = timed(in_tf), timed(gemm), timed(out_tf)
        t_dir = timed(lambda: F.conv2d(x, w16, padding=1))
        gf = 2.0 * B * H * H * K * C * 9 / 1e9
        # check against the direct conv
        Y = out_tf().reshape(B, th, th, 2, 2, K).permute(0, 5, 1, 3, 2, 4).reshape(B, K, H, H).float()
        ref = F.conv2d(x.float(), w16.float(), padding=1)
        err = float((Y - ref).abs().max() / ref.abs().max())
        print("%s %s: direct conv (torch / MIOpen) %.1f us = %.0f TFLOP/s | Winograd unfused: input transform %.1f + batched GEMM %.1f (%.0f TFLOP/s of its own %.1f GFLOP) "
              "+ output transform %.1f = %.1f us (%.2fx the algorithmic rate of torch's direct conv); max rel err vs direct %.1e"
              % (name, str(dt).split(".")[-1], t_dir, gf / t_dir * 1e3, t_in, t_g, gf / 2.25 / t_g * 1e3, gf / 2.25, t_out, t_in + t_g + t_out,
                 t_dir / (t_in + t_g + t_out), err))


if __name__ == "__main__":
    main()
