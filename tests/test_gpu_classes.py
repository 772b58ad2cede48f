"""The nets at the class counts the reference's drivers actually build (round 5; VERDICT r04 "missing" item 3):
VID's 31 classes in the TRN drivers (evaluate_trn.py:526-543, test_video_trn.py:34-56: ssd4scale_vgg static + temporal) and
COCO's 81 in evaluate_coco.py:245-270 (dualrefinedet_vggbn).  Every other net-level test runs the VOC count, 21.

What changes with the class count: the conf heads have 3 * C output channels (93 / 243 instead of 63), the deformable heads
12 + 3 * C columns (105 / 255) -- more than the 80 columns of a transform-then-sample Y row, so the 16-bit single-group plans
run 2 / 4 column groups (net.hip `y_groups`) -- softmax rows of C, Detect over C - 1 classes."""
import numpy as np
import pytest
import torch

from oracle import net_ref
from oracle import oracle as orc
from tdrn_amd import _lib
from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.utils import synth

import test_gpu_net as tgn
import test_gpu_pin16 as pin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("classes", [81, 31])
def test_drn_vggbn_fp32_matches_oracle_at_coco_and_vid_class_counts(classes):
    """evaluate_coco.py:262-264: build_net('test', 320, 81, c7_channel, def_groups, multihead, bn); every stage, the heads, Detect."""
    net, sd = tgn._build("dualrefinedet_vggbn", (320, classes, 1024, 1, True, True))
    x = synth.synth_frames(1, 320, seed=15)
    taps = {}
    ref_arm, ref_off, ref_odm, ref_conf = net_ref.drn_vggbn_forward(sd, x, classes, True, True, taps=taps)
    arm, offs, odm, conf = net(torch.from_numpy(x).to(DEV))
    rep = tgn._stage_report(net, 1, taps)
    assert len(rep) >= 30
    bad = [(l, e, m) for l, e, m in rep if e > 1e-3 * max(1.0, m)]
    assert not bad, "first diverging stages: %r" % bad[:5]
    assert conf.shape == (6375, classes) and odm.shape == (1, 6375, 4)
    np.testing.assert_allclose(arm.cpu().numpy(), ref_arm.numpy(), atol=1e-3, rtol=0)
    for a, b in zip(offs, ref_off):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    allowed = net_ref.border_rows(taps, True)
    tgn._close_mod_border_flips(odm.cpu().numpy(), ref_odm.numpy(), allowed)
    tgn._close_mod_border_flips(conf.cpu().numpy(), ref_conf.numpy(), allowed)
    assert torch.allclose(conf.sum(1), torch.ones_like(conf[:, 0]), atol=1e-5)
    # Detect over C - 1 classes: keep lists bit-exact against the oracle's Detect on the same fp32 inputs
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(classes, 0, 200, 0.01, 0.45).forward(odm, conf, pri, arm_loc_data=arm, scale=torch.tensor([500.0, 375.0, 500.0, 375.0])).cpu().numpy()
    mine = orc.detect(odm.cpu().numpy(), conf.cpu().numpy(), pri.cpu().numpy(), arm.cpu().numpy(), (500, 375, 500, 375), num_classes=classes)
    assert det.shape == (1, classes, 200, 5) and np.array_equal(det[..., 0], mine[..., 0])
    assert (det[0, 1:, 0, 0] > 0).any()


def test_trn_nets_fp32_match_oracle_at_31_classes():
    """evaluate_trn.py:541-543 on VID (31 classes): static net -> loc maps -> temporal net (8 deformable groups) -> the next frame
    on the cached offsets, all against the oracle run the reference's way."""
    C = 31
    stat, sd_s = tgn._build("ssd4scale_vgg", (320, C, 1024, True, False), seed=0)
    temp, sd_t = tgn._build("ssd4scale_vgg", (320, C, 1024, True, True), seed=1)
    clip = synth.synth_frames(2, 320, seed=41)
    xs = torch.from_numpy(clip).to(DEV)
    loc, conf, maps = stat(xs[:1], ret_loc=True)
    r_loc, r_conf, r_maps = net_ref.ssd4scale_vgg_forward(sd_s, clip[:1], C, "test", True, False, ret_loc=True)
    assert conf.shape == (6375, C)
    np.testing.assert_allclose(loc.cpu().numpy(), r_loc.numpy(), atol=1e-3, rtol=0)
    np.testing.assert_allclose(conf.cpu().numpy(), r_conf.numpy().reshape(conf.shape), atol=1e-3, rtol=0)
    for a, b in zip(maps, r_maps):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    tloc, tconf, offs = temp(xs[:1], ref_loc=maps, ret_off=True)
    rt_loc, rt_conf, r_offs = net_ref.ssd4scale_vgg_forward(sd_t, clip[:1], C, "test", True, True, ref_loc=r_maps, ret_off=True)
    for a, b in zip(offs, r_offs):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    np.testing.assert_allclose(tloc.cpu().numpy(), rt_loc.numpy(), atol=2e-3, rtol=0)
    np.testing.assert_allclose(tconf.cpu().numpy(), rt_conf.numpy().reshape(tconf.shape), atol=1e-3, rtol=0)
    o1 = temp(xs[1:2], offset_list=offs)
    r1 = net_ref.ssd4scale_vgg_forward(sd_t, clip[1:2], C, "test", True, True, offset_list=r_offs)
    np.testing.assert_allclose(o1[0].cpu().numpy(), r1[0].numpy(), atol=2e-3, rtol=0)
    np.testing.assert_allclose(o1[1].cpu().numpy(), r1[1].numpy().reshape(o1[1].shape), atol=1e-3, rtol=0)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(C, 0, 200, 0.01, 0.45).forward(tloc, tconf, pri, arm_loc_data=loc, scale=[500.0, 375.0, 500.0, 375.0]).cpu().numpy()
    mine = orc.detect(tloc.cpu().numpy(), tconf.cpu().numpy(), pri.cpu().numpy(), loc.cpu().numpy(), (500, 375, 500, 375), num_classes=C)
    assert np.array_equal(det[..., 0], mine[..., 0])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("classes", [81, 31])
def test_every_stage_from_its_own_input_other_class_counts(classes, dtype):
    """tests/test_gpu_pin16.py (b) + (c) on dualrefinedet_vggbn at 81 / 31 classes: every launch of the 16-bit plan from its own
    materialised input -- in particular the transform-then-sample heads in 4 / 2 column groups: each group's Y against an fp64
    GEMM with its rounded weights, the sampled outputs against a blend of the device's own Y rows."""
    net, sd = pin._build("dualrefinedet_vggbn", (320, classes, 1024, 1, True, True), phase="train", dtype=dtype)     # (raw logits)
    x = torch.from_numpy(synth.synth_frames(2, 320, seed=27)).to(DEV)
    report, checked = pin.check_stages(net, sd, x, dtype, images=(1,))
    ops = net._engine.op_infos()
    groups = [o["y_groups"] for o in ops if o["kind"] == "deform_heads"]
    assert groups == [(12 + 3 * classes + 79) // 80] * 4 and all(o["y"] >= 0 for o in ops if o["kind"] == "deform_heads")
    assert checked.get("deform_heads", 0) == 4 and checked.get("conv", 0) >= 30
    pin._print_report("dualrefinedet_vggbn %d classes %s" % (classes, dtype), report, checked)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_every_stage_trn_nets_31_classes(dtype):
    """the static and the temporal ssd4scale_vgg net at 31 classes, every conv launch from its own input (the grouped deformable
    heads of the temporal net run the exact gather kernel: covered by the fp32 test above and tests/test_gpu_ops.py)"""
    for deform in (False, True):
        net, sd = pin._build("ssd4scale_vgg", (320, 31, 1024, True, deform), phase="train", seed=int(deform), dtype=dtype)
        x = torch.from_numpy(synth.synth_frames(2, 320, seed=28)).to(DEV)
        if deform:
            stat, _ = tgn._build("ssd4scale_vgg", (320, 31, 1024, True, False), seed=0)
            maps = stat(x, ret_loc=True)[2]
            net_fwd = lambda xx: net(xx, ref_loc=maps)
        else:
            net_fwd = net
        report, checked = pin.check_stages(net, sd, x, dtype, images=(0,), forward=net_fwd)
        assert checked.get("conv", 0) >= (15 if deform else 20)


@pytest.mark.parametrize("classes", [31, 81])
def test_column_groups_agree_with_the_gather_kernel(classes):
    """16-bit plan with the transform-then-sample heads in column groups against the SAME 16-bit plan on the fused gather kernel
    (TDRN_PLAN_NO_DEFORM_TS): identical trunk bits, so the heads' inputs are identical and the two differ only by where the
    16-bit rounding sits (Y rounded vs. blended columns rounded): a few ulp16 of the head's magnitude, no column mixed up."""
    x = torch.from_numpy(synth.synth_frames(2, 320, seed=33)).to(DEV)
    outs = []
    for flags in (0, _lib.PLAN_NO_DEFORM_TS):
        net, sd = pin._build("dualrefinedet_vggbn", (320, classes, 1024, 1, True, True), phase="train", dtype="fp16", flags=flags)
        o = net(x)
        outs.append((o[0].clone(), o[2].clone(), o[3].clone()))
    assert torch.equal(outs[0][0], outs[1][0])                        # arm_loc: upstream of the heads
    for k in (1, 2):
        a, b = outs[0][k], outs[1][k]
        d = (a - b).abs()
        # (rows at a sampling discontinuity see the same offsets in both plans: nothing flips between them)
        assert float(d.max()) < 2e-2 * max(1.0, float(b.abs().max())), (k, float(d.max()))
        assert float(d.mean()) < 1e-3
