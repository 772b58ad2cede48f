"""Which stage's 16-bit rounding moves the detections?  (run on the GPU box.)

    python scripts/attribution.py [--dtype bf16] [--frames 8] [--out profiles/r04_attribution]

Method.  Two handles of dualrefinedet_vggbn-320 (multihead) with the same one-stream, two-launch-first-conv plan: one fp32 (its
outputs equal the CPU oracle's to 1e-5, tests/test_gpu_net.py), one in the 16-bit type.  For a boundary s of the plan, every
tensor that ops [0, s) produce is taken from the fp32 handle -- rounded ONCE to the 16-bit type by tdrn_net_write_tensor -- and
only the ops [s, end) run in 16 bits (tdrn_net_forward_from).  boxes(s) = two-stage decode of that run; s = 0 is the plain
16-bit forward, s = end the fp32 result.  The error that remains at boundary s is what the stages from s on contribute; the
difference between consecutive boundaries is the contribution of the stage in between (to first order: the map is not linear, a
tap that crosses a bilinear cell or the border of the map moves a box by a lot or not at all).

Writes  attribution_<dtype>.csv / .md : per boundary: mean and max error of the 50 boxes the plain 16-bit forward moves most,
                                        mean error of all boxes, and the detection agreement of Detect on the run's outputs with
                                        Detect on the fp32 outputs (matched at IoU >= 0.9 of rows_fp32)."""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tdrn_amd import _lib
from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.layers.box_utils import center_size, decode
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth

DEV = torch.device("cuda:0")


def make(dtype, sd):
    net = build_net("test", 320, 21, 1024, 1, True, True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    net.set_plan_flags(_lib.PLAN_ONE_STREAM | _lib.PLAN_NO_FUSE_FIRST)
    net.set_compute_dtype(dtype)
    return net, net.engine(DEV)


def boxes_of(r, pri):
    out = []
    for b in range(r["arm_loc"].shape[0]):
        out.append(decode(r["odm_loc"][b], center_size(decode(r["arm_loc"][b], pri, [0.1, 0.2])), [0.1, 0.2]))
    return torch.stack(out, 0)


def agreement(ref, got):
    """rows of `ref` (B, C, top_k, 5) matched one-to-one in `got` at IoU >= 0.9 inside each (image, class)"""
    ref, got = ref.cpu().numpy(), got.cpu().numpy()
    n_ref = matched = 0
    for i in range(ref.shape[0]):
        for c in range(1, ref.shape[1]):
            a, b = ref[i, c], got[i, c]
            ka, kb = a[:, 0] > 0, b[:, 0] > 0
            n_ref += int(ka.sum())
            if not ka.any() or not kb.any():
                continue
            ba, bb = a[ka][:, 1:], b[kb][:, 1:]
            iw = np.clip(np.minimum(ba[:, None, 2], bb[None, :, 2]) - np.maximum(ba[:, None, 0], bb[None, :, 0]), 0, None)
            ih = np.clip(np.minimum(ba[:, None, 3], bb[None, :, 3]) - np.maximum(ba[:, None, 1], bb[None, :, 1]), 0, None)
            inter = iw * ih
            iou = inter / np.maximum(((ba[:, 2] - ba[:, 0]) * (ba[:, 3] - ba[:, 1]))[:, None] + ((bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1]))[None, :] - inter, 1e-12)
            used = np.zeros(len(bb), bool)
            for j in range(len(ba)):
                k = int(np.argmax(np.where(used, -1.0, iou[j])))
                if not used[k] and iou[j, k] >= 0.9:
                    used[k] = True
                    matched += 1
    return n_ref, matched


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--out", default="profiles/r04_attribution")
    ap.add_argument("--reverse", action="store_true",
                    help="the other direction: ops [0, s) run in the 16-bit type, ops [s, end) in fp32 on the 16-bit run's tensors -- "
                         "the best ANY mixed plan with a 16-bit prefix could do (round 5)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    net0 = build_net("test", 320, 21, 1024, 1, True, True)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net0.state_dict().items()}, 0)
    n32, e32 = make("fp32", sd)
    n16, e16 = make(args.dtype, sd)
    B = args.frames
    x = torch.from_numpy(synth.synth_frames(B, 320, seed=77)).to(DEV)          # the frames of bench.py's detection agreement
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45)
    scale = torch.tensor([500.0, 375.0, 500.0, 375.0])

    r32 = e32.forward(x, want_offsets=True)
    torch.cuda.synchronize()
    ops32, ops16 = e32.op_infos(), e16.op_infos()
    assert [(o["kind"], o["w"]) for o in ops32] == [(o["kind"], o["w"]) for o in ops16], "the two plans differ"
    t32 = {}
    for o in ops32:
        for t in (o["out"], o["pool"]):
            if t >= 0 and e32.tensor_infos()[t][0]:
                t32[t] = e32.read_tensor(t, B).clone()
    box32 = boxes_of(r32, pri)
    det32 = det.forward(r32["odm_loc"], r32["conf"], pri, arm_loc_data=r32["arm_loc"], scale=scale).clone()

    def run_from(s):
        out = {"arm_loc": r32["arm_loc"].clone(), "odm_loc": r32["odm_loc"].clone(), "conf": r32["conf"].clone()}
        if s > 0:
            e16.workspace(B)
            names16 = e16.tensor_infos()
            for j, o in enumerate(ops16[:s]):
                for t in (o["out"], o["pool"]):
                    if t >= 0 and names16[t][0] and t in t32:
                        e16.write_tensor(t, t32[t])
        r = e16.forward(x, out=out, first_op=s)
        torch.cuda.synchronize()
        return r

    if args.reverse:
        return reverse(args, e32, e16, ops16, x, B, pri, det, scale, r32, box32, det32)

    r16 = run_from(0)
    e0 = (boxes_of(r16, pri) - box32).abs().amax(-1)                           # (B, P): worst coordinate of every box
    worst = torch.topk(e0.flatten(), 50).indices
    n_ref, m0 = agreement(det32, det.forward(r16["odm_loc"], r16["conf"], pri, arm_loc_data=r16["arm_loc"], scale=scale))
    rows = []
    boundaries = [i for i, o in enumerate(ops16) if o["kind"] in ("conv", "conv_transpose", "l2norm", "maxpool", "offset_conv", "deform_heads", "first_conv")]
    first_deform = min(i for i, o in enumerate(ops16) if o["kind"] == "deform_heads")
    boundaries = [b for b in boundaries if b <= first_deform]                  # (the four deformable ops are one launch)
    prev = None
    for s in boundaries:
        r = run_from(s)
        e = (boxes_of(r, pri) - box32).abs().amax(-1)
        ew = e.flatten()[worst]
        _, m = agreement(det32, det.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale))
        o = ops16[s]
        name = "%s:%s" % (o["kind"], o["w"] or e16.tensor_infos()[o["in"]][0])
        row = dict(first_op=s, first_16bit_stage=name, worst50_mean=float(ew.mean()), worst50_max=float(ew.max()), all_mean=float(e.mean()),
                   all_max=float(e.max()), boxes_over_1e3=int((e > 1e-3).sum()), detections_matched=m, detections_fp32=n_ref)
        rows.append(row)
        print("%3d %-36s worst50 mean %.4f max %.4f | all mean %.2e max %.3f, %5d boxes > 1e-3 | detections matched %d / %d" % (
            s, name, row["worst50_mean"], row["worst50_max"], row["all_mean"], row["all_max"], row["boxes_over_1e3"], m, n_ref), flush=True)
    # contribution of the stage BETWEEN two boundaries = drop of the remaining error
    for i, row in enumerate(rows):
        nxt = rows[i + 1] if i + 1 < len(rows) else None
        row["stage_contribution_worst50_mean"] = row["worst50_mean"] - (nxt["worst50_mean"] if nxt else 0.0)
        row["stage_contribution_all_mean"] = row["all_mean"] - (nxt["all_mean"] if nxt else 0.0)
    keys = list(rows[0].keys())
    with open(os.path.join(args.out, "attribution_%s.csv" % args.dtype), "w") as f:
        w = csv.DictWriter(f, fieldnames=keys)
        w.writeheader()
        w.writerows(rows)
    with open(os.path.join(args.out, "attribution_%s.md" % args.dtype), "w") as f:
        f.write("# %s: what remains of the box error when the plan runs in fp32 up to a boundary (dualrefinedet_vggbn 320, %d frames, %d priors each)\n\n" % (args.dtype, B, box32.shape[1]))
        f.write("Boxes in normalised image coordinates; `worst50` = the 50 boxes the plain %s forward moves most (max %.3f).\n" % (args.dtype, float(e0.max())))
        f.write("`contribution` = error remaining when the 16-bit part starts AT this stage minus when it starts at the next one.\n\n")
        f.write("| first 16-bit stage | worst50 mean | worst50 max | contribution (worst50 mean) | all boxes mean | contribution (all mean) | boxes > 1e-3 | detections matched |\n|---|---|---|---|---|---|---|---|\n")
        for r_ in rows:
            f.write("| %s | %.4f | %.4f | %+.4f | %.2e | %+.2e | %d | %d / %d |\n" % (r_["first_16bit_stage"], r_["worst50_mean"], r_["worst50_max"], r_["stage_contribution_worst50_mean"],
                                                                                  r_["all_mean"], r_["stage_contribution_all_mean"], r_["boxes_over_1e3"], r_["detections_matched"], r_["detections_fp32"]))
    print("plain %s forward: detections matched %d / %d" % (args.dtype, m0, n_ref))


def reverse(args, e32, e16, ops16, x, B, pri, det, scale, r32, box32, det32):
    """16-bit prefix, fp32 suffix: what would a mixed plan buy whose LAST stages run exactly?"""
    r16 = e16.forward(x, want_offsets=True)
    torch.cuda.synchronize()
    names16 = e16.tensor_infos()
    t16 = {}
    for o in ops16:
        for t in (o["out"], o["pool"]):
            if t >= 0 and names16[t][0]:
                t16[t] = e16.read_tensor(t, B).clone()
    e0 = (boxes_of(r16, pri) - box32).abs().amax(-1)
    _, m0 = agreement(det32, det.forward(r16["odm_loc"], r16["conf"], pri, arm_loc_data=r16["arm_loc"], scale=scale))
    boundaries = [i for i, o in enumerate(ops16) if o["kind"] in ("conv", "conv_transpose", "offset_conv", "deform_heads")]
    first_deform = min(i for i, o in enumerate(ops16) if o["kind"] == "deform_heads")
    boundaries = [b for b in boundaries if b <= first_deform and b > 0]
    rows = [dict(first_fp32_stage="(none: plain %s)" % args.dtype, all_mean=float(e0.mean()), all_max=float(e0.max()),
                 boxes_over_1e3=int((e0 > 1e-3).sum()), boxes_over_1e2=int((e0 > 1e-2).sum()), detections_matched=m0)]
    names32 = e32.tensor_infos()
    for s in boundaries:
        out = {"arm_loc": r16["arm_loc"].clone(), "odm_loc": r16["odm_loc"].clone(), "conf": r16["conf"].clone()}
        e32.workspace(B)
        for o in ops16[:s]:
            for t in (o["out"], o["pool"]):
                if t >= 0 and names32[t][0] and t in t16:
                    e32.write_tensor(t, t16[t])
        r = e32.forward(x, out=out, first_op=s)
        torch.cuda.synchronize()
        e = (boxes_of(r, pri) - box32).abs().amax(-1)
        _, m = agreement(det32, det.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale))
        o = ops16[s]
        name = "%s:%s" % (o["kind"], o["w"] or names16[o["in"]][0])
        rows.append(dict(first_fp32_stage=name, all_mean=float(e.mean()), all_max=float(e.max()), boxes_over_1e3=int((e > 1e-3).sum()),
                         boxes_over_1e2=int((e > 1e-2).sum()), detections_matched=m))
        print("%3d %-36s all mean %.2e max %.3f, %5d boxes > 1e-3, %4d > 1e-2 | detections matched %d" % (
            s, name, rows[-1]["all_mean"], rows[-1]["all_max"], rows[-1]["boxes_over_1e3"], rows[-1]["boxes_over_1e2"], m), flush=True)
    with open(os.path.join(args.out, "reverse_%s.md" % args.dtype), "w") as f:
        f.write("# %s prefix, fp32 suffix: box error left when ops [0, s) run in %s and ops [s, end) in fp32 on the %s run's tensors\n\n" % (args.dtype, args.dtype, args.dtype))
        f.write("dualrefinedet_vggbn 320, %d frames, %d priors each; boxes in normalised image coordinates; detections of 32 000.\n" % (B, box32.shape[1]))
        f.write("The upper bound of what ANY mixed plan with a 16-bit trunk can reach: the suffix is exact, its inputs carry the prefix's drift.\n\n")
        f.write("| first fp32 stage | all boxes mean | max | boxes > 1e-3 | boxes > 1e-2 | detections matched |\n|---|---|---|---|---|---|\n")
        for r_ in rows:
            f.write("| %s | %.2e | %.4f | %d | %d | %d |\n" % (r_["first_fp32_stage"], r_["all_mean"], r_["all_max"], r_["boxes_over_1e3"], r_["boxes_over_1e2"], r_["detections_matched"]))


if __name__ == "__main__":
    main()
