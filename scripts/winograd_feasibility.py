"""Round 6, review item 6: Winograd F(2x2, 3x3) feasibility for the 3x3 trunk layers -- the NUMERICS half, on the host.

For conv3_2 and conv4_2 of dualrefinedet_vggbn (synthetic weights of bench.py, the net's OWN activations from the fp32 oracle):
the layer is recomputed in fp64 (the yardstick) and then the way each device variant would compute it --
  direct bf16 / fp16 : inputs and BN-folded weights rounded to the type, exact products, wide accumulation, one rounding of the output
                       (what conv3x3_patch / conv3x3_pp do; tests/test_gpu_pin16.py holds them to this model);
  Winograd fp16 / bf16: V = B^T d B of the 16-bit input tile in fp32, rounded ONCE to the type; U = G g G^T of the folded weights in
                       fp64, rounded once; M = sum_c U V exact products / wide accumulation (the matrix cores); Y = A^T M A + bias in fp32;
                       one rounding of the output.
Printed per layer: error of every variant against fp64 (max and RMS, absolute and relative to the RMS of the layer's output), the
largest |V| (fp16 overflows at 65504) -- the numbers DESIGN.md quotes next to the cost model that fails the speed gate.
This script is an analysis tool: it imports the oracle (test infrastructure) and no product code except the synthetic-weight helper."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import net_ref                                   # noqa: E402
from tdrn_amd.utils import synth                             # noqa: E402


def round_bf16(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def round_f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


ROUND = {"bf16": round_bf16, "fp16": round_f16}

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def winograd(x16, w64, bias64, rnd):
    """x16: (C, H, W) float32 holding 16-bit values; w64: (K, C, 3, 3) float64 BN-folded; returns (K, H, W) float32 before the output rounding."""
    C, H, W = x16.shape
    K = w64.shape[0]
    xp = np.zeros((C, H + 2, W + 2), dtype=np.float32)
    xp[:, 1:-1, 1:-1] = x16
    th, tw = H // 2, W // 2
    # 4x4 input tiles, stride 2: d[c, ty, tx, 4, 4]
    d = np.empty((C, th, tw, 4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            d[:, :, :, i, j] = xp[:, i:i + 2 * th:2, j:j + 2 * tw:2]
    bt32 = BT.astype(np.float32)
    V = np.einsum("ai,ctxij,bj->ctxab", bt32, d, bt32, optimize=True).astype(np.float32)     # fp32 adds of 16-bit values: exact here
    vmax = float(np.abs(V).max())
    V = rnd(V)
    U = rnd(np.einsum("ai,kcij,bj->kcab", G, w64, G, optimize=True))                        # (K, C, 4, 4)
    M = np.einsum("kcab,ctxab->ktxab", U.astype(np.float64), V.astype(np.float64), optimize=True).astype(np.float32)
    Y = np.einsum("pa,ktxab,qb->ktxpq", AT.astype(np.float32), M, AT.astype(np.float32), optimize=True).astype(np.float32)
    out = Y.transpose(0, 1, 3, 2, 4).reshape(K, H, W) + bias64.astype(np.float32)[:, None, None]
    return out.astype(np.float32), vmax


def main():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    net = build_net("test", 320, 21, 1024, 1, True, True)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    x = torch.from_numpy(synth.synth_frames(1, 320, seed=100))
    taps = {}
    net_ref.vgg_trunk(sdt, x, True, taps)
    print("layer      variant          max|err|    rms err   rms err / rms(y)   max|err| / max|y|   max|V|")
    for name, src, conv, bn in (("conv3_2", "backbone.14", "backbone.17", "backbone.18"), ("conv4_2", "backbone.24", "backbone.27", "backbone.28")):
        xin = taps[src][0].numpy()                                                        # (C, H, W) fp32, post-ReLU
        w = sd[conv + ".weight"].astype(np.float64)
        b = sd[conv + ".bias"].astype(np.float64)
        g, beta, mean, var = (sd[bn + s].astype(np.float64) for s in (".weight", ".bias", ".running_mean", ".running_var"))
        sc = g / np.sqrt(var + 1e-5)
        w64 = w * sc[:, None, None, None]
        b64 = (b - mean) * sc + beta
        yref = F.conv2d(torch.from_numpy(xin.astype(np.float64))[None], torch.from_numpy(w64), torch.from_numpy(b64), padding=1)[0].numpy()
        yref = np.maximum(yref, 0.0)
        rms_y, max_y = float(np.sqrt((yref ** 2).mean())), float(np.abs(yref).max())
        for dt in ("bf16", "fp16"):
            rnd = ROUND[dt]
            x16 = rnd(xin)
            yd = F.conv2d(torch.from_numpy(x16.astype(np.float64))[None], torch.from_numpy(rnd(w64).astype(np.float64)), torch.from_numpy(b64), padding=1)[0].numpy()
            yd = rnd(np.maximum(yd, 0.0).astype(np.float32))
            yw, vmax = winograd(x16, w64, b64, rnd)
            yw = rnd(np.maximum(yw, 0.0))
            for label, y, vm in (("direct " + dt, yd, None), ("winograd " + dt, yw, vmax)):
                e = y.astype(np.float64) - yref
                print("%-10s %-14s %10.3e %10.3e %14.3e %18.3e %s" % (name, label, np.abs(e).max(), np.sqrt((e ** 2).mean()),
                      np.sqrt((e ** 2).mean()) / rms_y, np.abs(e).max() / max_y, "" if vm is None else "%10.1f" % vm))


if __name__ == "__main__":
    main()
