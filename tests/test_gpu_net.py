"""GPU parity of the whole network path (build_net -> net(x) -> Detect) against the CPU oracle and
the golden fixtures made from the reference's own forward."""
import os

import numpy as np
import pytest
import torch

from oracle import net_ref
from oracle import oracle as orc
from tdrn_amd import _lib
from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(modname, args, seed=0):
    import importlib
    net = importlib.import_module("tdrn_amd.model." + modname).build_net("test", *args)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net.to(DEV), sd


def _close_mod_border_flips(got, ref, allowed, atol=1e-3, width=None):
    """fp32 parity of an output that sits BEHIND the deformable heads.  The sampling rule is discontinuous where a
    sample coordinate crosses 0 or the map size (deform_conv_cuda_kernel.cu:195: -1e-7 samples 0, +1e-7 the border
    pixel), so an offset that differs in the last bit -- any change of summation order upstream does that -- can
    flip one output pixel by O(1).  `allowed` (oracle.net_ref.border_rows, from the ORACLE's offsets) marks the
    prior rows of pixels with a tap within 1e-4 of such a discontinuity: every row over `atol` must be one of
    them, so any other 6-row bug fails.  The offsets themselves are compared strictly by the callers."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    width = width or got.shape[-1]
    bad = (np.abs(got - ref) > atol).reshape(-1, width).any(axis=1)
    allowed = np.asarray(allowed, bool).reshape(-1)
    assert bad.shape == allowed.shape
    stray = bad & ~allowed
    assert not stray.any(), "%d rows differ by more than %g away from any sampling discontinuity (first %r, max %g)" % (
        int(stray.sum()), atol, np.nonzero(stray)[0][:5].tolist(), float(np.abs(got - ref).reshape(-1, width)[stray].max()))
    assert int(allowed.sum()) <= 30, "suspiciously many pixels on a discontinuity: %d rows" % int(allowed.sum())


def _stage_report(net, B, taps):
    """max abs error of every internal activation the oracle also exposes (localises a failure)."""
    eng = net._engine
    rep = []
    for i, (label, c, h, w) in enumerate(eng.tensor_infos()):
        if label in taps:
            got = eng.read_tensor(i, B).cpu().numpy()
            ref = taps[label].numpy()
            if got.shape == ref.shape:
                rep.append((label, float(np.abs(got - ref).max()), float(np.abs(ref).max())))
    return rep


@pytest.mark.parametrize("mh", [True, False], ids=["multihead", "singlehead"])
def test_drn_vggbn_fp32_matches_oracle_every_stage(mh):
    net, sd = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, mh))
    x = synth.synth_frames(2, 320, seed=5)
    taps = {}
    ref_arm, ref_off, ref_odm, ref_conf = net_ref.drn_vggbn_forward(sd, x, 21, True, mh, taps=taps)
    arm, offs, odm, conf = net(torch.from_numpy(x).to(DEV))
    rep = _stage_report(net, 2, taps)
    assert len(rep) >= 30
    bad = [(l, e, m) for l, e, m in rep if e > 1e-3 * max(1.0, m)]
    assert not bad, "first diverging stages: %r" % bad[:5]
    assert arm.shape == (2, 6375, 4) and odm.shape == (2, 6375, 4) and conf.shape == (2 * 6375, 21)
    assert [tuple(o.shape) for o in offs] == [(2, 18, f, f) for f in (40, 20, 10, 5)]
    np.testing.assert_allclose(arm.cpu().numpy(), ref_arm.numpy(), atol=1e-3, rtol=0)
    for a, b in zip(offs, ref_off):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    allowed = net_ref.border_rows(taps, mh)
    _close_mod_border_flips(odm.cpu().numpy(), ref_odm.numpy(), allowed)
    _close_mod_border_flips(conf.cpu().numpy(), ref_conf.numpy(), allowed)
    assert torch.allclose(conf.sum(1), torch.ones_like(conf[:, 0]), atol=1e-5)


@pytest.mark.parametrize("tag,mh", [("drn_vggbn_320_mh", True), ("drn_vggbn_320", False)])
def test_drn_vggbn_matches_reference_golden(golden_dir, tag, mh):
    """Fixture = the reference's own RefineSSD.forward + Detect (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    net, sd = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, mh))
    x = torch.from_numpy(synth.synth_frames(1, 320, 0)).to(DEV)
    arm, offs, odm, conf = net(x)
    sub = int(g["sub"])
    taps = {}
    net_ref.drn_vggbn_forward(sd, synth.synth_frames(1, 320, 0), 21, True, mh, taps=taps)     # (the fixture's inputs)
    allowed = net_ref.border_rows(taps, mh)[::sub]
    np.testing.assert_allclose(arm.cpu().numpy()[:, ::sub], g["arm_loc"], atol=1e-3, rtol=0)
    _close_mod_border_flips(odm.cpu().numpy()[:, ::sub], g["odm_loc"], allowed)
    _close_mod_border_flips(conf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["conf"], allowed)
    np.testing.assert_allclose(offs[3].cpu().numpy(), g["off3"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(offs[0].cpu().numpy()[:, :, ::5, ::5], g["off0"], atol=1e-3, rtol=0)
    # evaluate.py protocol: Detect on the net's output; compare with the reference's Detect output.
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45).forward(odm, conf, pri, arm_loc_data=arm,
                                                  scale=torch.tensor([500.0, 375.0, 500.0, 375.0])).cpu().numpy()
    ref = g["detect"]
    # NMS decisions can flip where the fp32 paths differ by 1e-3: require the bulk to agree exactly
    same_rows = (np.abs(det - ref).max(-1) < 2e-3).mean()
    assert same_rows > 0.97, same_rows
    # and bit-exact keep lists when the oracle's Detect is fed OUR net outputs (identical fp32 inputs)
    mine = orc.detect(odm.cpu().numpy(), conf.cpu().numpy(), pri.cpu().numpy(), arm.cpu().numpy(),
                      (500, 375, 500, 375))
    assert np.array_equal(det[..., 0], mine[..., 0])
    np.testing.assert_allclose(det, mine, rtol=3e-6, atol=1e-6)


# Measured drift of the 16-bit modes vs the fp32 oracle (scripts/drift_table.py on frame seed 5 -> profiles/r02_drift/families.csv):
# (mean over everything, 99.9th percentile over the prior rows AWAY from a sampling discontinuity).  The deformable
# sampling rule is discontinuous at the map border (deform_conv_cuda_kernel.cu:195), so pixels with a tap within
# NEAR_EPS of it -- 2.5x the measured offset drift of the dtype -- flip by O(1) in ANY reduced precision; they are
# excluded from the percentile (not from the mean) and counted.  The tests allow 1.5x these numbers.  The error
# grows smoothly with depth (profiles/r02_drift/stages_*.csv: 0.14 % after conv1_1 to 1.1 % at the end of the FPN in
# bf16, 8x less in fp16): it is the input rounding of every layer, not one bad kernel.
NEAR_EPS = {"bf16": 0.03, "fp16": 0.004}
DRIFT_MEASURED = {
    ("dualrefinedet_vggbn", "bf16"): {"arm": (2.397e-03, 4.897e-02), "odm": (1.633e-02, 1.070e-01), "conf": (8.883e-04, 3.278e-02)},
    ("dualrefinedet_vggbn", "fp16"): {"arm": (2.951e-04, 5.226e-03), "odm": (2.153e-03, 1.202e-02), "conf": (1.209e-04, 3.931e-03)},
    ("dualrefinedet_mobilenet", "bf16"): {"arm": (4.875e-03, 1.401e-01), "odm": (5.922e-02, 7.513e-01), "conf": (1.555e-03, 1.139e-01)},
    ("dualrefinedet_mobilenet", "fp16"): {"arm": (6.273e-04, 1.974e-02), "odm": (7.578e-03, 8.582e-02), "conf": (2.143e-04, 1.584e-02)},
    ("refinedet_vgg", "bf16"): {"arm": (2.397e-03, 4.897e-02), "odm": (1.905e-02, 9.033e-02), "conf": (8.866e-04, 3.654e-02)},
    ("refinedet_vgg", "fp16"): {"arm": (2.951e-04, 5.226e-03), "odm": (2.461e-03, 1.183e-02), "conf": (1.130e-04, 4.646e-03)},
    ("ssd4scale_vgg", "bf16"): {"arm": (2.397e-03, 4.897e-02), "conf": (1.055e-04, 4.957e-03)},
    ("ssd4scale_vgg", "fp16"): {"arm": (2.951e-04, 5.226e-03), "conf": (1.305e-05, 6.488e-04)},
    ("ssd4scale_mobile", "bf16"): {"arm": (4.875e-03, 1.401e-01), "conf": (1.643e-04, 1.598e-02)},
    ("ssd4scale_mobile", "fp16"): {"arm": (6.273e-04, 1.974e-02), "conf": (2.048e-05, 1.960e-03)},
}
FAMILIES = {   # build_net args (after phase) and the oracle forward -> (arm_loc | loc, odm_loc | None, conf)
    "dualrefinedet_vggbn": ((320, 21, 1024, 1, True, True), lambda sd, x, taps: net_ref.drn_vggbn_forward(sd, x, 21, True, True, taps=taps)),
    "dualrefinedet_mobilenet": ((320, 21, 1, True), lambda sd, x, taps: net_ref.drn_mobilenet_forward(sd, x, 21, True, taps=taps)),
    "refinedet_vgg": ((320, 21, True, 1024, True, True), lambda sd, x, taps: net_ref.refinedet_vgg_forward(sd, x, 21, True, True, True)),
    "ssd4scale_vgg": ((320, 21, 1024, True, False), lambda sd, x, taps: net_ref.ssd4scale_vgg_forward(sd, x, 21, "test", True)),
    "ssd4scale_mobile": ((320, 21, 1024, False), lambda sd, x, taps: net_ref.ssd4scale_mobile_forward(sd, x, 21, "test")),
}


def _split_outputs(o):
    return (o[0], o[2], o[3]) if len(o) == 4 else (o[0], None, o[1])


def _check_drift(family, dtype, net, sd, x):
    """Runs `net` (already in `dtype`) on frame(s) x and checks its drift against 1.5x DRIFT_MEASURED[(family, dtype)]."""
    taps = {}
    r_arm, r_odm, r_conf = _split_outputs(FAMILIES[family][1](sd, x, taps))
    arm, odm, conf = _split_outputs(net(torch.from_numpy(x).to(DEV)))
    near = net_ref.border_rows(taps, True, NEAR_EPS[dtype]) if taps else None
    rep = {}
    for name, got, ref in (("arm", arm, r_arm), ("odm", odm, r_odm), ("conf", conf, r_conf)):
        if got is None:
            continue
        e = (got.cpu() - ref.reshape(got.shape)).abs()
        away = e.reshape(-1, e.shape[-1])
        if near is not None and name != "arm":
            away = away[torch.from_numpy(~near)]
        rep[name] = (float(e.mean()), float(torch.quantile(away.flatten()[:: max(1, away.numel() // 4000000)], 0.999)))
    print("%s %s drift (mean, p99.9 away from discontinuities): %r; rows near: %d" % (family, dtype, rep, 0 if near is None else int(near.sum())))
    for name, (m, q) in DRIFT_MEASURED[(family, dtype)].items():
        assert rep[name][0] < 1.5 * m and rep[name][1] < 1.5 * q, (family, dtype, name, rep[name], (m, q))
    if near is not None:
        assert near.mean() < 0.25


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_16bit_drift_is_bounded(family, dtype):
    """bf16/fp16 drift of every model family is reported separately from the 1e-3 fp32 claim and held
    to 1.5x the committed measurement."""
    net, sd = _build(family, FAMILIES[family][0])
    net.set_compute_dtype(dtype)
    _check_drift(family, dtype, net, sd, synth.synth_frames(1, 320, seed=5))


def test_16bit_error_scales_with_the_significand():
    """Two checks of the 16-bit modes against something OUTSIDE this repo's own measurements (DRIFT_MEASURED is a regression
    guard):
    (1) analytic, first layer: conv1_1 computes sum(x*w') with x and the BN-folded w' rounded to the 16-bit type (relative error
        <= u = 2^-8 bf16 / 2^-11 fp16 each), fp32 accumulation, one rounding of the output -- so every element must satisfy
        |y16 - y32| <= (2u + u^2) S + u (|y32| + (2u + u^2) S) + 1e-6 S,   S = sum |x| |w'|  (ReLU is 1-Lipschitz);
    (2) end to end: the error of a whole net is the accumulated input rounding of its layers, so fp16 (11-bit significand)
        must be at least 5x closer to the fp32 oracle than bf16 (8 bits; ideal ratio 8) on every output of every family."""
    import torch.nn.functional as F
    x = synth.synth_frames(1, 320, seed=5)
    # (1)
    for dtype, u in (("bf16", 2.0 ** -8), ("fp16", 2.0 ** -11)):
        net, sd = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
        net.set_plan_flags(_lib.PLAN_NO_FUSE_FIRST)             # (conv1_1's output is a tensor of its own in this plan)
        net.set_compute_dtype(dtype)
        net(torch.from_numpy(x).to(DEV))
        eng = net._engine
        idx = [i for i, (lab, c, h, w) in enumerate(eng.tensor_infos()) if lab == "backbone.0"][0]
        got = eng.read_tensor(idx, 1).cpu().double()
        w = torch.from_numpy(sd["backbone.0.weight"]).double()
        sc = torch.from_numpy(sd["backbone.1.weight"]).double() / torch.sqrt(torch.from_numpy(sd["backbone.1.running_var"]).double() + 1e-5)
        wf = w * sc[:, None, None, None]
        bf = (torch.from_numpy(sd["backbone.0.bias"]).double() - torch.from_numpy(sd["backbone.1.running_mean"]).double()) * sc \
            + torch.from_numpy(sd["backbone.1.bias"]).double()
        xt = torch.from_numpy(x).double()
        y = F.conv2d(xt, wf, bf, padding=1)
        S = F.conv2d(xt.abs(), wf.abs(), None, padding=1)
        e1 = (2 * u + u * u) * S
        bound = e1 + u * (y.abs() + e1) + 1e-6 * S
        err = (got - y.clamp(min=0)).abs()
        assert bool((err <= bound).all()), (dtype, float((err - bound).max()))
        assert float(err.mean()) > 0.02 * float(bound.mean())        # (the bound is not vacuous: within 50x of the observed error)
    # (2)
    for family in sorted(FAMILIES):
        errs = {}
        for dtype in ("bf16", "fp16"):
            net, sd = _build(family, FAMILIES[family][0])
            net.set_compute_dtype(dtype)
            ref = _split_outputs(FAMILIES[family][1](sd, x, {}))
            got = _split_outputs(net(torch.from_numpy(x).to(DEV)))
            errs[dtype] = [float((g.cpu() - r.reshape(g.shape)).abs().mean()) for g, r in zip(got, ref) if g is not None]
        for eb, eh in zip(errs["bf16"], errs["fp16"]):
            assert eh * 5.0 <= eb, (family, errs)


def test_batch32_rows_equal_single_frame_runs():
    """Full BASELINE batch (32 frames, bf16): every frame's outputs are bit-identical to running it
    alone -- tile boundaries and batch position must not leak into the arithmetic."""
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
    net.bfloat16()
    x = torch.from_numpy(synth.synth_frames(32, 320, seed=7)).to(DEV)
    arm, offs, odm, conf = net(x)
    conf = conf.view(32, 6375, 21)
    assert torch.isfinite(arm).all() and torch.isfinite(odm).all() and torch.isfinite(conf).all()
    for b in (0, 17, 31):
        a1, o1, d1, c1 = net(x[b:b + 1])
        assert torch.equal(a1[0], arm[b]) and torch.equal(d1[0], odm[b]) and torch.equal(c1, conf[b])
        assert torch.equal(o1[2][0], offs[2][b])


def test_batch192_rows_equal_single_frame_runs():
    """Six times the batch the plan was made for (split-K and batch-minor tiles are planned at batch 32; at 192 the small
    maps have enough tiles that a split-K slice of a border tile can have no live tap at all -- conv_igemm.hip's
    3-stage path once went into its epilogue with that slice's LDS-DMA still in flight): every checked
    frame equals its single-frame run bit for bit, in both 16-bit types."""
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
    x = torch.from_numpy(synth.synth_frames(32, 320, seed=8)).to(DEV).repeat(6, 1, 1, 1)
    x[100] = x[100].flip(-1)
    for dtype in ("bf16", "fp16"):
        net.set_compute_dtype(dtype)
        arm, offs, odm, conf = net(x)
        conf = conf.view(192, 6375, 21)
        for b in (0, 100, 191):
            a1, o1, d1, c1 = net(x[b:b + 1])
            assert torch.equal(a1[0], arm[b]) and torch.equal(d1[0], odm[b]) and torch.equal(c1, conf[b])
            assert torch.equal(o1[0][0], offs[0][b]) and torch.equal(o1[3][0], offs[3][b])


def test_drn_mobilenet_fp32_matches_oracle():
    net, sd = _build("dualrefinedet_mobilenet", (320, 21, 1, True))
    x = synth.synth_frames(2, 320, seed=11)
    taps = {}
    ref_arm, _, ref_odm, ref_conf = net_ref.drn_mobilenet_forward(sd, x, 21, True, taps=taps)
    arm, none, odm, conf = net(torch.from_numpy(x).to(DEV))
    assert none is None
    allowed = net_ref.border_rows(taps, True)
    np.testing.assert_allclose(arm.cpu().numpy(), ref_arm.numpy(), atol=1e-3, rtol=0)
    _close_mod_border_flips(odm.cpu().numpy(), ref_odm.numpy(), allowed)
    _close_mod_border_flips(conf.cpu().numpy(), ref_conf.numpy(), allowed)


@pytest.mark.parametrize("tag,mh", [("mh", True), ("sh", False)])
def test_drn_mobilenet_matches_reference_golden(golden_dir, tag, mh):
    """BASELINE config #4's model against the reference's own forward (tests/golden/make_golden.py,
    model/dualrefinedet_mobilenet.py:127-199), multihead on and off."""
    g = np.load(os.path.join(golden_dir, "drn_mobilenet_320.npz"))
    sub, seed = int(g["sub"]), int(g["x_seed"])
    net, sd = _build("dualrefinedet_mobilenet", (320, 21, 1, mh))
    x = synth.synth_frames(1, 320, seed)
    arm, none, odm, conf = net(torch.from_numpy(x).to(DEV))
    assert none is None
    taps = {}
    net_ref.drn_mobilenet_forward(sd, x, 21, mh, taps=taps)
    allowed = net_ref.border_rows(taps, mh)[::sub]
    np.testing.assert_allclose(arm.cpu().numpy()[:, ::sub], g[tag + "_arm"], atol=1e-3, rtol=0)
    _close_mod_border_flips(odm.cpu().numpy()[:, ::sub], g[tag + "_odm"], allowed, atol=2e-3)     # |odm| ~ 2.6 here
    _close_mod_border_flips(conf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g[tag + "_conf"], allowed)


def test_ssd4scale_mobile_static_and_temporal_nets():
    """Config #1 model, and the TRN key-frame protocol (evaluate_trn.py:438-467): the static net's
    raw loc maps drive the temporal net's 8-group deformable heads."""
    stat, sd_s = _build("ssd4scale_mobile", (320, 21, 1024, False), seed=0)
    temp, sd_t = _build("ssd4scale_mobile", (320, 21, 1024, True), seed=1)
    x = synth.synth_frames(2, 320, seed=13)
    xs = torch.from_numpy(x).to(DEV)
    loc, conf, loc_maps = stat(xs, ret_loc=True)
    r_loc, r_conf, r_maps = net_ref.ssd4scale_mobile_forward(sd_s, x, 21, "test", False, ret_loc=True)
    np.testing.assert_allclose(loc.cpu().numpy(), r_loc.numpy(), atol=1e-3, rtol=0)
    np.testing.assert_allclose(conf.cpu().numpy(), r_conf.numpy(), atol=1e-3, rtol=0)
    for a, b in zip(loc_maps, r_maps):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    assert len(stat(xs)) == 2
    t_loc, t_conf, offs = temp(xs, ref_loc=loc_maps, ret_off=True)
    rt_loc, rt_conf, r_offs = net_ref.ssd4scale_mobile_forward(sd_t, x, 21, "test", True, ref_loc=r_maps, ret_off=True)
    for a, b in zip(offs, r_offs):
        assert tuple(a.shape) == tuple(b.shape)
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    np.testing.assert_allclose(t_loc.cpu().numpy(), rt_loc.numpy(), atol=2e-3, rtol=0)
    np.testing.assert_allclose(t_conf.cpu().numpy(), rt_conf.numpy(), atol=1e-3, rtol=0)
    # cached offsets (non key frames): same result as recomputing them
    t2 = temp(xs, offset_list=offs)
    assert torch.equal(t2[0], t_loc) and torch.equal(t2[1], t_conf)
    # ssd4scale feeds Detect without ARM refinement (evaluate.py:457-458)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45).forward(loc, conf, pri).cpu().numpy()
    ref = orc.detect(loc.cpu().numpy(), conf.cpu().numpy(), pri.cpu().numpy(), None, (320,) * 4)
    assert np.array_equal(det[..., 0], ref[..., 0])


def _rows_equal_single_runs(net, x, rows, unpack=_split_outputs):
    full = unpack(net(x))
    B = x.size(0)
    for t in full:
        assert t is None or bool(torch.isfinite(t).all())
    for b in rows:
        one = unpack(net(x[b:b + 1]))
        for f, o in zip(full, one):
            if f is None:
                continue
            f = f.view(B, -1, f.shape[-1])
            assert torch.equal(f[b], o.view(1, -1, o.shape[-1])[0]), "batch row %d differs from the single-frame run" % b
    return full


def test_config3_vggbn_512_fp16_batch16():
    """BASELINE config #3 at its stated size: 512x512, fp16, batch 16, deformable path on.  Every frame's outputs
    are bit-identical to running it alone; Detect on the batch equals the oracle's Detect on the same outputs."""
    net, sd = _build("dualrefinedet_vggbn", (512, 21, 1024, 1, True, True))
    net.half()
    x = torch.from_numpy(synth.synth_frames(16, 512, seed=33)).to(DEV)
    arm, odm, conf = _rows_equal_single_runs(net, x, (0, 9, 15))
    assert arm.shape == (16, 16320, 4) and conf.shape == (16 * 16320, 21)
    pri = PriorBox(mb_cfg["VOC_512_RefineDet"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45).forward(odm, conf, pri, arm_loc_data=arm).cpu().numpy()
    b = 11
    mine = orc.detect(odm[b:b + 1].cpu().numpy(), conf.view(16, 16320, 21)[b].cpu().numpy(), pri.cpu().numpy(),
                      arm[b:b + 1].cpu().numpy(), (320,) * 4)
    assert np.array_equal(det[b][..., 0], mine[0][..., 0])


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_config4_drn_mobilenet_batch64(dtype):
    """BASELINE config #4's per-GPU workload: dualrefinedet_mobilenet 320, batch 64.  The config names no dtype: the
    deployment default for such configs is fp16 (same rate as bf16, 8x less drift: DESIGN.md "Which 16-bit type");
    bf16 is run as well (the depthwise strip kernel in both 16-bit forms).  Batch rows == single-frame runs bit for
    bit, drift of one frame within the table."""
    net, sd = _build("dualrefinedet_mobilenet", (320, 21, 1, True))
    net.set_compute_dtype(dtype)
    x = torch.from_numpy(synth.synth_frames(64, 320, seed=35)).to(DEV)
    # (rows 7, 15, 63: frames whose last pixel tiles are TAIL SUB-ITEMS of pw1x1_kernel at this batch -- 800 items of a 512 -> 512 layer
    # are 3.125 rounds; an XCD's last four items run as sixteen 64-cout quarter items, round 6 -- against conv_igemm at batch 1)
    arm, odm, conf = _rows_equal_single_runs(net, x, (0, 7, 15, 33, 63))
    assert arm.shape == (64, 6375, 4) and conf.shape == (64 * 6375, 21)
    _check_drift("dualrefinedet_mobilenet", dtype, net, sd, synth.synth_frames(1, 320, seed=5))
    # whole items only: the same bits
    other, _ = _build("dualrefinedet_mobilenet", (320, 21, 1, True))
    other.set_compute_dtype(dtype)
    other.set_plan_flags(_lib.PLAN_NO_PATCH_TAIL)
    for u, v in zip(_split_outputs(other(x)), (arm, odm, conf)):
        assert u is None or torch.equal(u, v)


@pytest.mark.parametrize("dtype16", ["fp16", "bf16"])
def test_config5_trn_8_clips_of_4_frames(dtype16):
    """(the config names no dtype: fp16 is the deployment default, bf16 runs too)  BASELINE config #5: the TRN temporal path on 8 clips x 4 frames (evaluate_trn.py:438-467 batched over clips):
    key frame -> static net (loc maps) -> temporal net (offsets), the three following frames reuse the cached
    offsets.  Every clip's results are bit-identical to running that clip alone; the fp32 key-frame pass of one
    clip matches the oracle.  refinedet_vgg (the model the config names) at batch 8 likewise."""
    stat, sd_s = _build("ssd4scale_vgg", (320, 21, 1024, True, False), seed=0)
    temp, sd_t = _build("ssd4scale_vgg", (320, 21, 1024, True, True), seed=1)
    clips = torch.from_numpy(synth.synth_frames(32, 320, seed=37)).to(DEV).view(8, 4, 3, 320, 320)

    def run(cl, dtype):
        for n in (stat, temp):
            if n.compute_dtype != dtype:
                n.set_compute_dtype(dtype)
        s_loc, s_conf, maps = stat(cl[:, 0].contiguous(), ret_loc=True)
        outs = [temp(cl[:, 0].contiguous(), ref_loc=maps, ret_off=True)]
        for f in range(1, 4):
            outs.append(temp(cl[:, f].contiguous(), offset_list=outs[0][2]))
        return s_loc, maps, outs
    s_loc, maps, outs = run(clips, dtype16)
    for c in (0, 5, 7):
        s1, m1, o1 = run(clips[c:c + 1], dtype16)
        assert torch.equal(s1[0], s_loc[c]) and torch.equal(m1[0][0], maps[0][c])
        for f in range(4):
            assert torch.equal(o1[f][0][0], outs[f][0][c]), (c, f)
            assert torch.equal(o1[f][1], outs[f][1].view(8, 6375, 21)[c])
        assert torch.equal(o1[0][2][1][0], outs[0][2][1][c])
    # fp32, one clip, against the oracle
    c = 3
    s1, m1, o1 = run(clips[c:c + 1], "fp32")
    xk = clips[c, 0:1].cpu().numpy()
    r_loc, r_conf, r_maps = net_ref.ssd4scale_vgg_forward(sd_s, xk, 21, "test", True, False, ret_loc=True)
    np.testing.assert_allclose(s1.cpu().numpy(), r_loc.numpy(), atol=1e-3, rtol=0)
    rt_loc, rt_conf, r_offs = net_ref.ssd4scale_vgg_forward(sd_t, xk, 21, "test", True, True, ref_loc=r_maps, ret_off=True)
    np.testing.assert_allclose(o1[0][0].cpu().numpy(), rt_loc.numpy(), atol=2e-3, rtol=0)
    np.testing.assert_allclose(o1[0][1].cpu().numpy(), rt_conf.numpy(), atol=1e-3, rtol=0)
    x1 = clips[c, 1:2].cpu().numpy()
    r1 = net_ref.ssd4scale_vgg_forward(sd_t, x1, 21, "test", True, True, offset_list=r_offs)
    np.testing.assert_allclose(o1[1][0].cpu().numpy(), r1[0].numpy(), atol=2e-3, rtol=0)
    # refinedet_vgg, batch 8
    rd, _ = _build("refinedet_vgg", (320, 21, True, 1024, True, True))
    rd.set_compute_dtype(dtype16)
    _rows_equal_single_runs(rd, clips[:, 0].contiguous(), (0, 4, 7))


def test_state_dict_roundtrip_and_missing_num_batches_tracked():
    net, sd = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, False))
    x = torch.from_numpy(synth.synth_frames(1, 320, seed=2)).to(DEV)
    a = net(x)
    old = {k: torch.from_numpy(v) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    net.load_state_dict(old)                                  # PyTorch-0.4 style checkpoint
    b = net(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
    with pytest.raises(ValueError):
        net(x[:, :, :300, :300])


def test_refinedet_vgg_matches_oracle_and_reference(golden_dir):
    """model/refinedet_vgg.py (config #5 names it): 4-tuple with use_refine (+bn +multihead: the 3x3+5x5 heads
    are packed as one merged 5x5 conv) and the plain 2-tuple variant, vs the reference's own CPU forward."""
    g = np.load(os.path.join(golden_dir, "other_models.npz"))
    sub = int(g["sub"])
    x = synth.synth_frames(1, 320, 21)
    net, sd = _build("refinedet_vgg", (320, 21, True, 1024, True, True))
    arm, none, odm, conf = net(torch.from_numpy(x).to(DEV))
    assert none is None
    np.testing.assert_allclose(arm.cpu().numpy()[:, ::sub], g["rd_arm"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(odm.cpu().numpy()[:, ::sub], g["rd_odm"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(conf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["rd_conf"], atol=1e-3, rtol=0)
    r = net_ref.refinedet_vgg_forward(sd, x, 21, True, True, True)
    np.testing.assert_allclose(odm.cpu().numpy(), r[2].numpy(), atol=1e-3, rtol=0)
    net, sd = _build("refinedet_vgg", (320, 21, False, 1024, False, False))
    out = net(torch.from_numpy(x).to(DEV))
    assert len(out) == 2
    np.testing.assert_allclose(out[0].cpu().numpy()[:, ::sub], g["rd0_odm"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(out[1].cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["rd0_conf"], atol=1e-3, rtol=0)


def test_ssd4scale_vgg_trn_protocol_matches_reference(golden_dir):
    """The nets the TRN drivers actually build (evaluate_trn.py:541-543): static -> loc maps -> temporal."""
    g = np.load(os.path.join(golden_dir, "other_models.npz"))
    sub = int(g["sub"])
    xs = torch.from_numpy(synth.synth_frames(1, 320, 21)).to(DEV)
    stat, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, False), seed=0)
    temp, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, True), seed=1)
    loc, conf, maps = stat(xs, ret_loc=True)
    np.testing.assert_allclose(loc.cpu().numpy()[:, ::sub], g["sv_loc"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(conf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["sv_conf"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(maps[3].cpu().numpy(), g["sv_map3"], atol=1e-3, rtol=0)
    tloc, tconf, offs = temp(xs, ref_loc=maps, ret_off=True)
    np.testing.assert_allclose(offs[3].cpu().numpy(), g["sv_off3"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(tloc.cpu().numpy()[:, ::sub], g["sv_tloc"], atol=2e-3, rtol=0)
    np.testing.assert_allclose(tconf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["sv_tconf"], atol=1e-3, rtol=0)
    # clip of 4 frames at interval 4 (config #5): key frame once, cached offsets afterwards
    clip = torch.from_numpy(synth.synth_frames(4, 320, 22)).to(DEV)
    s_out = stat(clip[:1], ret_loc=True)
    first = temp(clip[:1], ref_loc=s_out[2], ret_off=True)
    for f in range(1, 4):
        o = temp(clip[f:f + 1], offset_list=first[2])
        assert o[0].shape == (1, 6375, 4) and torch.isfinite(o[1]).all()


def test_ssd4scale_mobile_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "other_models.npz"))
    sub = int(g["sub"])
    xs = torch.from_numpy(synth.synth_frames(1, 320, 21)).to(DEV)
    stat, _ = _build("ssd4scale_mobile", (320, 21, 1024, False), seed=0)
    temp, _ = _build("ssd4scale_mobile", (320, 21, 1024, True), seed=1)
    loc, conf, maps = stat(xs, ret_loc=True)
    np.testing.assert_allclose(loc.cpu().numpy()[:, ::sub], g["sm_loc"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(conf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["sm_conf"], atol=1e-3, rtol=0)
    tloc, tconf, offs = temp(xs, ref_loc=maps, ret_off=True)
    np.testing.assert_allclose(tloc.cpu().numpy()[:, ::sub], g["sm_tloc"], atol=2e-3, rtol=0)
    np.testing.assert_allclose(tconf.cpu().numpy().reshape(1, -1, 21)[:, ::sub], g["sm_tconf"], atol=1e-3, rtol=0)


def test_drn_vggbn_512_fp32_and_fp16():
    """BASELINE config #3 geometry: 512x512, P = 16320, deformable path on; fp32 vs the oracle, fp16 drift."""
    net, sd = _build("dualrefinedet_vggbn", (512, 21, 1024, 1, True, True))
    x = synth.synth_frames(1, 512, seed=31)
    taps = {}
    r_arm, r_off, r_odm, r_conf = net_ref.drn_vggbn_forward(sd, x, 21, True, True, taps=taps)
    arm, offs, odm, conf = net(torch.from_numpy(x).to(DEV))
    assert arm.shape == (1, 16320, 4) and conf.shape == (16320, 21)
    assert [tuple(o.shape) for o in offs] == [(1, 18, f, f) for f in (64, 32, 16, 8)]
    allowed = net_ref.border_rows(taps, True)
    np.testing.assert_allclose(arm.cpu().numpy(), r_arm.numpy(), atol=1e-3, rtol=0)
    _close_mod_border_flips(odm.cpu().numpy(), r_odm.numpy(), allowed)
    _close_mod_border_flips(conf.cpu().numpy(), r_conf.numpy(), allowed)
    pri = PriorBox(mb_cfg["VOC_512_RefineDet"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45).forward(odm, conf, pri, arm_loc_data=arm).cpu().numpy()   # default scale [320]*4
    mine = orc.detect(odm.cpu().numpy(), conf.cpu().numpy(), pri.cpu().numpy(), arm.cpu().numpy(), (320,) * 4)
    assert np.array_equal(det[..., 0], mine[..., 0])
    net.half()
    a16, _, o16, c16 = net(torch.from_numpy(synth.synth_frames(2, 512, seed=31)).to(DEV))
    e = (o16[:1].cpu() - r_odm).abs().flatten()
    assert float(e.mean()) < 0.006 and float(torch.quantile(e[::7], 0.999)) < 0.06
    assert float((c16[:16320].cpu() - r_conf).abs().mean()) < 6e-4


def test_trn_driver_key_frame_protocol():
    """evaluate_trn.py:438-467: static net on key frames only, cached offsets in between, anchors from the
    static net's loc; the driver's detections equal the hand-written sequence of calls."""
    from tdrn_amd.trn import TRNDriver
    stat, _ = _build("ssd4scale_mobile", (320, 21, 1024, False), seed=0)
    temp, _ = _build("ssd4scale_mobile", (320, 21, 1024, True), seed=1)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.3, 0.45)
    drv = TRNDriver(stat, temp, det, pri, interval=4, loose=1.0, deform=True)
    clip = torch.from_numpy(synth.synth_frames(6, 320, seed=41)).to(DEV)
    outs = [drv.step(clip[f:f + 1], video_name="v0") for f in range(6)]
    assert drv.key_frames == 2                      # frames 0 and 4
    s0 = stat(clip[0:1], ret_loc=True)
    t0 = temp(clip[0:1], ref_loc=s0[2], ret_off=True)
    t2 = temp(clip[2:3], offset_list=t0[2])
    ref2 = det.forward(t2[0], t2[1], pri, arm_loc_data=s0[0])
    assert torch.equal(outs[2], ref2)
    s4 = stat(clip[4:5], ret_loc=True)
    t5 = temp(clip[5:6], ref_loc=s4[2])
    assert torch.equal(outs[5], det.forward(t5[0], t5[1], pri, arm_loc_data=s4[0]))
    drv.step(clip[0:1], video_name="v1")            # a new video forces a key frame
    assert drv.key_frames == 3 and drv.current_i == 1
    # whole intervals of several clips at once (TRNDriver.clips): the same detections as the frame-by-frame protocol
    two = torch.from_numpy(synth.synth_frames(8, 320, seed=43)).to(DEV).view(2, 4, 3, 320, 320)          # (clip, frame, ...)
    want = []
    for c in range(2):
        d2 = TRNDriver(stat, temp, det, pri, interval=4, loose=1.0, deform=True)
        want.append([d2.step(two[c, f:f + 1], video_name="c%d" % c).clone() for f in range(4)])
    got = drv.clips(two.transpose(0, 1).contiguous())                                                   # frame-major
    for f in range(4):
        for c in range(2):
            assert torch.equal(got[f * 2 + c], want[c][f][0]), (f, c)
    # ... with the static net on a second stream beside the temporal trunk: the same detections
    side = torch.cuda.Stream(DEV)
    again = drv.clips(two.transpose(0, 1).contiguous(), side_stream=side)
    torch.cuda.synchronize()
    assert torch.equal(again, got)


@pytest.mark.parametrize("size,mh", [(192, False), (384, True)])
def test_net_runs_other_input_sizes_like_the_reference(size, mh):
    """multi_eval.py:526-547 feeds a 320-net frames of 192..704 pixels (the nets are fully convolutional): a
    second plan over the SAME packed weights, fp32 parity with the oracle at that size."""
    net, sd = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, mh))
    x = synth.synth_frames(1, size, seed=17)
    taps = {}
    r_arm, r_off, r_odm, r_conf = net_ref.drn_vggbn_forward(sd, x, 21, True, mh, taps=taps)
    net(torch.from_numpy(synth.synth_frames(1, 320, seed=17)).to(DEV))      # the 320 plan packs the weights
    arm, offs, odm, conf = net(torch.from_numpy(x).to(DEV))
    P = 3 * sum((size // s) ** 2 for s in (8, 16, 32, 64))
    assert arm.shape == (1, P, 4) and conf.shape == (P, 21)
    assert net.engine_for(torch.from_numpy(x).to(DEV)).weights.data_ptr() == net._engine.weights.data_ptr()
    np.testing.assert_allclose(arm.cpu().numpy(), r_arm.numpy(), atol=1e-3, rtol=0)
    for a, b in zip(offs, r_off):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-3, rtol=0)
    allowed = net_ref.border_rows(taps, mh)
    _close_mod_border_flips(odm.cpu().numpy(), r_odm.numpy(), allowed)
    _close_mod_border_flips(conf.cpu().numpy(), r_conf.numpy(), allowed)
    if size == 384:
        # shape_check "input image is smaller than kernel" (deform_conv_cuda.c:75): 5x5 heads on the 3x3 map of 192
        with pytest.raises(_lib.TdrnError):
            net(torch.from_numpy(synth.synth_frames(1, 192, seed=17)).to(DEV))


def test_multi_scale_flip_tester():
    """multi_eval.py:513-631 end to end on the device for scales 192 + 320 of a 320-net: the un-flipped 320 view
    is exactly the plain pipeline, every view fills its key, and the merged result is bbox_vote of the parts."""
    from tdrn_amd.data import base_transform, multi_cfg, MEANS
    from tdrn_amd.eval import MultiScaleTester, merge_detections
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, False))
    rng = np.random.Generator(np.random.PCG64(9))
    frame = torch.from_numpy(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)).to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45)
    pri = {s: PriorBox(multi_cfg[str(s)]).forward().to(DEV) for s in (192, 320)}
    tester = MultiScaleTester(net, det, pri, ssd_dim=320, mean=MEANS, scales=[192, 320])
    voted, multi = tester.detect(frame)
    assert sorted(multi) == ["320_192_0", "320_192_1", "320_320_0", "320_320_1"]
    assert all(v.shape == (1, 21, 200, 5) for v in multi.values())
    arm, _, odm, conf = net(base_transform(frame, 320, MEANS, True))
    plain = det.forward(odm, conf, pri[320], arm_loc_data=arm).cpu().numpy()
    assert np.array_equal(multi["320_320_0"], plain)
    again = merge_detections(multi, 400, 300, 320, 21)
    assert sorted(voted) == sorted(again) and all(np.array_equal(voted[j], again[j]) for j in voted)
    for j, boxes in voted.items():
        assert boxes.shape[1] == 5 and (boxes[:, 4] > 0).all()


@pytest.mark.parametrize("ssd_dim,mh", [(320, False), (512, True)])
def test_multi_scale_tester_runs_the_references_full_scale_lists(ssd_dim, mh):
    """multi_eval.py:21-24: a 320-net is tested at [192, 320, 384, 448, 512, 576, 704] and a 512-net at [320, 512, 640,
    1216] (P up to 30855 / 92055: beyond what an LDS-resident Detect could hold).  Every view yields a full result;
    at the largest scale the device pipeline equals the oracle's Detect on the same net outputs, bit for bit."""
    from tdrn_amd.data import base_transform, multi_cfg, multi_cfg_512, MEANS
    from tdrn_amd.eval import MultiScaleTester
    from tdrn_amd.eval.tta import MULTI_SCALE
    net, _ = _build("dualrefinedet_vggbn", (ssd_dim, 21, 1024, 1, True, mh))
    cfgs = multi_cfg if ssd_dim == 320 else multi_cfg_512
    scales = MULTI_SCALE[str(ssd_dim)]
    assert scales == ([192, 320, 384, 448, 512, 576, 704] if ssd_dim == 320 else [320, 512, 640, 1216])
    pri = {s: PriorBox(cfgs[str(s)]).forward().to(DEV) for s in scales}
    rng = np.random.Generator(np.random.PCG64(19))
    frame = torch.from_numpy(rng.integers(0, 256, (360, 480, 3), dtype=np.uint8)).to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45)
    tester = MultiScaleTester(net, det, pri, ssd_dim=ssd_dim, mean=MEANS)
    voted, multi = tester.detect(frame)
    assert sorted(multi) == sorted("%d_%d_%d" % (ssd_dim, s, f) for s in scales for f in (0, 1))
    assert all(v.shape == (1, 21, 200, 5) and np.isfinite(v).all() for v in multi.values())
    assert len(voted) > 0
    top = scales[-1]
    P = pri[top].shape[0]
    assert P == 3 * sum((top // st) ** 2 for st in (8, 16, 32, 64)) and P > 16384
    x = base_transform(frame, top, MEANS, True)
    arm, _, odm, conf = net(x)
    got = det.forward(odm, conf, pri[top], arm_loc_data=arm).cpu().numpy()
    assert np.array_equal(got, multi["%d_%d_0" % (ssd_dim, top)])
    # the oracle's full cpu_nms over ~P candidates x 20 classes is slow at P = 92055: check five classes there
    classes = range(1, 21) if P < 40000 else (1, 5, 9, 14, 20)
    cf = conf.cpu().numpy().copy()
    keep_cols = np.zeros(21, bool)
    keep_cols[list(classes)] = True
    cf[:, ~keep_cols] = 0.0                                     # (below conf_thresh: those classes produce no candidates)
    ref = orc.detect(odm.cpu().numpy(), cf, pri[top].cpu().numpy(), arm.cpu().numpy(), (320,) * 4)
    for c in classes:
        assert np.array_equal(got[0, c, :, 0], ref[0, c, :, 0]), c
        np.testing.assert_allclose(got[0, c], ref[0, c], rtol=3e-6, atol=1e-6)


def test_hipgraph_replay_equals_eager():
    """The whole step (net forward on its four stream lanes + Detect) captured as ONE hipGraph (engine.GraphedCall):
    replays are bit-identical to eager launches, also after the input buffer is refilled."""
    from tdrn_amd.engine import GraphedCall
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
    net.bfloat16()
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45)

    def step(x):
        arm, _, odm, conf = net(x)
        return arm, odm, conf, det.forward(odm, conf, pri, arm_loc_data=arm, scale=[500.0, 375.0, 500.0, 375.0])
    xa = torch.from_numpy(synth.synth_frames(4, 320, seed=61)).to(DEV)
    xb = torch.from_numpy(synth.synth_frames(4, 320, seed=62)).to(DEV)
    ea = [t.clone() for t in step(xa)]
    eb = [t.clone() for t in step(xb)]
    g = GraphedCall(step, xa)
    for x, e in ((xa, ea), (xb, eb), (xa, ea)):
        out = g(x)
        torch.cuda.synchronize()
        for got, want in zip(out, e):
            assert torch.equal(got, want)
    # one graph per shape: another batch size or dtype is refused, not silently run on stale buffers
    with pytest.raises(ValueError):
        g(xa[:2])
    with pytest.raises(ValueError):
        g(xa.double())
    # clone_outputs: a result kept across replays is not overwritten by the next one
    gc = GraphedCall(step, xa, clone_outputs=True)
    keep = gc(xa)
    gc(xb)
    torch.cuda.synchronize()
    for got, want in zip(keep, ea):
        assert torch.equal(got, want)


def test_fused_first_conv_at_a_batch_past_4gib_of_input():
    """The fused first conv never reads the (B,H,W,64) conv1_1 tensor, so the 2^32-byte input limit of the patch kernel does
    not apply to it (the planner decided from geometry only and a refused launch failed the whole forward):
    ssd4scale_vgg at 512 px, batch 136 (136*512*512*64*2 B = 4.25 GiB of conv1_1 output never materialised) runs, and its
    first frames equal a batch-2 run."""
    net, _ = _build("ssd4scale_vgg", (512, 21, 1024, False, False))
    net.set_compute_dtype("fp16")
    x = torch.from_numpy(synth.synth_frames(2, 512, seed=91)).to(DEV)
    big = x.repeat(68, 1, 1, 1)
    small = [t.clone() if torch.is_tensor(t) else t for t in net(x)]
    out = net(big)
    torch.cuda.synchronize()
    for u, v in zip(out, small):
        if torch.is_tensor(u) and torch.is_tensor(v) and u.dim() > 0 and u.shape[0] == 136:
            assert torch.equal(u[:2], v)
            assert torch.equal(u[134:], v)


@pytest.mark.parametrize("model,args", [("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True)), ("ssd4scale_vgg", (320, 21, 1024, False, False))])
def test_fused_first_conv_equals_two_launches(model, args, monkeypatch):
    """16-bit plans compute the first conv inside conv1_2's patch loader (conv3x3_patch.hip FUSE): same operand layout and
    instruction as the stand-alone kernel, so every output is BIT-identical to the two-launch plan (plan_flags TDRN_PLAN_NO_FUSE_FIRST) --
    at the build size, at other frame sizes (tiles at every border, fewer items than CUs) and at batch 1 / 3 / 8."""
    # (the multihead DRN needs a >= 5x5 coarsest map for its 5x5 deformable heads: sizes from 320 up)
    cases = [(320, 3), (320, 8), (384, 3), (448, 1), (704, 1)] + ([(192, 3), (256, 1)] if model == "ssd4scale_vgg" else [])
    monkeypatch.delenv("TDRN_FUSE_FIRST", raising=False)
    for dtype in ("bf16", "fp16"):
        fused, _ = _build(model, args)                       # two handles in one process, differing only in plan_flags
        fused.set_compute_dtype(dtype)
        plain, _ = _build(model, args)
        plain.set_plan_flags(_lib.PLAN_NO_FUSE_FIRST)
        plain.set_compute_dtype(dtype)
        for size, batch in cases:
            x = torch.from_numpy(synth.synth_frames(batch, size, seed=70 + size + batch)).to(DEV)
            a = fused(x)
            b = plain(x)
            for u, v in zip(a, b):
                if torch.is_tensor(u):
                    assert torch.equal(u, v), (model, dtype, size, batch)
                else:
                    for uu, vv in zip(u, v):
                        assert torch.equal(uu, vv), (model, dtype, size, batch)
        # (not vacuous: the two plans do differ)
        names = []
        for net in (fused, plain):
            eng = net._engine
            eng.set_profile(1)
            xs = torch.from_numpy(synth.synth_frames(1, 320, seed=3)).to(DEV)
            eng.forward(xs)
            torch.cuda.synchronize()
            names.append([o["name"].split(":")[0] for o in eng.op_stats()])
            eng.set_profile(0)
        assert "first_conv" not in names[0] and names[1][0] == "first_conv"


@pytest.mark.parametrize("model,args", [("dualrefinedet_mobilenet", (320, 21, 1, True)), ("ssd4scale_mobile", (320, 21, 1024, False))])
def test_fused_dwpw_equals_two_launches(model, args):
    """Opt-in plan TDRN_PLAN_DWPW: eight of the MobileNet trunk's conv_dw blocks (model/networks.py:736-745) as ONE launch each
    (dwpw.hip dwpw_kernel: the depthwise output stays in LDS; measured slower than the two launches, hence opt-in).  Same
    depthwise FMA order, same rounding of the intermediate, same K order and bias placement as the default plan (dwconv3_strip +
    the 1x1 GEMM): every output BIT-identical, at the build size and other frame sizes (2-D and flat tiles, ragged last tiles)
    and at batches 1 / 3 / 8."""
    cases = [(320, 1), (320, 3), (320, 8), (384, 2), (512, 1)] + ([(256, 3)] if model == "ssd4scale_mobile" else [])
    for dtype in ("bf16", "fp16"):
        fused, _ = _build(model, args)
        fused.set_plan_flags(_lib.PLAN_DWPW)
        fused.set_compute_dtype(dtype)
        plain, _ = _build(model, args)
        plain.set_compute_dtype(dtype)
        for size, batch in cases:
            x = torch.from_numpy(synth.synth_frames(batch, size, seed=90 + size + batch)).to(DEV)
            a = fused(x)
            b = plain(x)
            for u, v in zip(a, b):
                if torch.is_tensor(u):
                    assert torch.equal(u, v), (model, dtype, size, batch)
                elif u is not None:
                    for uu, vv in zip(u, v):
                        assert torch.equal(uu, vv), (model, dtype, size, batch)
        names = []
        for net in (fused, plain):                              # (not vacuous: the launch lists differ)
            eng = net._engine
            eng.set_profile(1)
            eng.forward(torch.from_numpy(synth.synth_frames(1, 320, seed=3)).to(DEV))
            torch.cuda.synchronize()
            names.append([o["name"].split(":")[0] for o in eng.op_stats()])
            eng.set_profile(0)
        assert names[0].count("dwpw_mfma") == 8 and "dwpw_mfma" not in names[1] and len(names[0]) == len(names[1]) - 8


def test_chain_launch_equals_one_launch_per_layer():
    """Opt-in plan TDRN_PLAN_CHAIN: the small top-of-pyramid layers (extras, last TCB level, its up-sampling, the lateral of the
    level below) as ONE launch -- a queue of tiles and split-K reduce ranges with per-stage completion counters
    (conv_igemm.hip conv_chain_kernel; measured slower than the launches it replaces, hence opt-in).  Same tile code, same K
    order, same split: every output is BIT-identical to the default plan, at batch 1 / 5 / 32, also when the inputs change
    between forwards (a stale read of the previous forward's activations would show) and under repetition."""
    model, args, size = "dualrefinedet_vggbn", (320, 21, 1024, 1, True, True), 320
    for dtype in ("bf16", "fp32"):
        chained, _ = _build(model, args)
        chained.set_plan_flags(_lib.PLAN_CHAIN)
        chained.set_compute_dtype(dtype)
        plain, _ = _build(model, args)
        plain.set_compute_dtype(dtype)
        for batch in ((1, 5, 32) if dtype != "fp32" else (2,)):
            for rep in range(3):
                x = torch.from_numpy(synth.synth_frames(batch, size, seed=100 + 7 * rep + batch)).to(DEV)
                a = chained(x)
                b = plain(x)
                for u, v in zip(a, b):
                    if torch.is_tensor(u):
                        assert torch.equal(u, v), (model, dtype, batch, rep)
                    else:
                        for uu, vv in zip(u, v):
                            assert torch.equal(uu, vv), (model, dtype, batch, rep)
        # (not vacuous: the chained plan has fewer launches, one of them the chain)
        names = []
        for net in (chained, plain):
            eng = net._engine
            eng.set_profile(1)
            eng.forward(torch.from_numpy(synth.synth_frames(1, size, seed=3)).to(DEV))
            torch.cuda.synchronize()
            names.append([o["name"].split(":")[0] for o in eng.op_stats()])
            eng.set_profile(0)
        assert names[0].count("conv_chain") == 1 and "conv_chain" not in names[1] and len(names[0]) < len(names[1]) - 2


def test_frame_stream_equals_unstreamed():
    """tdrn_amd.stream.FrameStream (copy-in stream two batches ahead | one hipGraph per slot: preprocess -> net -> Detect |
    copy-out stream, chained by events; test_video.py:98-115 as a pipeline) returns, batch after batch, exactly what the same
    step gives without capture, slots or copies -- also when a slot's buffers are re-used turn after turn and when the
    producer hands over each batch only just in time."""
    from tdrn_amd.stream import FrameStream
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
    net.set_compute_dtype("fp16")
    eng = net.engine(DEV)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    B, slots, n = 3, 3, 8
    fs = FrameStream(eng, Detect(21, 0, 200, 0.01, 0.45), pri, B, slots=slots)
    rng = np.random.RandomState(11)
    feeds = [torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3), dtype=np.uint8)) for _ in range(n)]
    fs.prime(feeds[:2])
    results, pending = [], []
    for k in range(n):                                     # batch k runs in slot k % 3; batch k+2 is handed over before the launch
        if k + 2 < n:
            assert fs.next_in() == (k + 2) % slots
            fs.pinned_in(fs.next_in()).copy_(feeds[k + 2])
        s = fs.run()
        assert s == k % slots
        pending.append(s)
        if len(pending) == 2:                              # read results one step late, as a consumer overlapping with the device would
            results.append(fs.result(pending.pop(0)).clone())
    while pending:
        results.append(fs.result(pending.pop(0)).clone())
    fs.drain()
    for k, got in enumerate(results):
        want = fs.eager(feeds[k].to(DEV)).cpu()
        assert torch.equal(got, want), k
    assert (results[0][..., 0] > 0).any()                  # (not vacuous: there are detections)
    with pytest.raises(ValueError):
        FrameStream(eng, Detect(21, 0, 200, 0.01, 0.45), pri, B, slots=2)
    # a consumer that lags by `slots` steps gets an error, not a newer batch's detections
    fs.prime(feeds[:2])
    old = fs.run()
    for _ in range(slots):
        last = fs.run()
    with pytest.raises(RuntimeError):
        fs.result(old)
    assert fs.result(last) is not None and int(last) == int(old)
    fs.drain()


@pytest.mark.parametrize("model,args", [("dualrefinedet_mobilenet", (320, 21, 1, True)), ("ssd4scale_mobile", (320, 21, 1024, False))])
def test_dw_sliding_window_equals_strip_kernel(model, args):
    """The depthwise 3x3 layers of the MobileNet trunks (conv_dw, model/networks.py:736-745) run on one of two kernels: a thread
    makes a 4-pixel strip of ONE output row (dwconv3_strip_kernel) or walks that strip down a segment of up to 8 rows keeping the
    input rows in registers (dwconv3_slide_kernel, chosen when the launch still fills the chip: the batch decides).  Same tap
    order per output, so the choice must not change a bit: plans that force either kernel and the default plan give identical
    outputs, at batches on both sides of the switch, stride 1 and 2, segments and strips that end ragged (20 / 10 / 5 / 3-row maps,
    24- and 12-pixel rows at 384), fp32 included."""
    cases = [(320, 1), (320, 3), (320, 40), (384, 2), (512, 1)] + ([(256, 3)] if model == "ssd4scale_mobile" else [])
    for dtype in ("bf16", "fp16", "fp32"):
        nets = []
        for flags in (0, _lib.PLAN_NO_DW_SLIDE, _lib.PLAN_DW_SLIDE_ALL):
            net, _ = _build(model, args)
            net.set_plan_flags(flags)
            net.set_compute_dtype(dtype)
            nets.append(net)
        for size, batch in (cases if dtype != "fp32" else cases[:2]):
            x = torch.from_numpy(synth.synth_frames(batch, size, seed=190 + size + batch)).to(DEV)
            outs = [net(x) for net in nets]
            for other in outs[1:]:
                for u, v in zip(outs[0], other):
                    if torch.is_tensor(u):
                        assert torch.equal(u, v), (model, dtype, size, batch)
                    elif u is not None:
                        for uu, vv in zip(u, v):
                            assert torch.equal(uu, vv), (model, dtype, size, batch)


@pytest.mark.parametrize("model,targs", [("ssd4scale_vgg", (320, 21, 1024, True)), ("ssd4scale_mobile", (320, 21, 1024))])
def test_trn_key_frame_broadcast_equals_the_frame_loop(model, targs):
    """tdrn_net_io.reserved[1]: ONE temporal forward over all frames of a step's clips (frame-major; ref_loc maps of the Bk key frames
    only, frame i uses the offsets of key frame i % Bk) against the reference's loop -- the key frame with ref_loc, every following
    frame of the interval with the cached offset_list (evaluate_trn.py:452-462).  The frames depend on the key frame only through
    those offsets, so every output is BIT-identical; so are the offsets handed out.  Also through the reuse token (a second batched
    forward with the offsets left in the workspace), and the argument checks."""
    for dtype in ("bf16", "fp32"):
        stat, sd_s = _build(model, targs + (False,), seed=0)
        temp, sd_t = _build(model, targs + (True,), seed=1)
        for n in (stat, temp):
            n.set_compute_dtype(dtype)
        Bk, F = (3, 4) if dtype == "bf16" else (2, 2)
        frames = torch.from_numpy(synth.synth_frames(F * Bk, 320, seed=41)).to(DEV).view(F, Bk, 3, 320, 320)      # frame-major
        _, _, maps = stat(frames[0], ret_loc=True)
        first = temp(frames[0], ref_loc=maps, ret_off=True)
        loop = [first] + [temp(frames[f], offset_list=first[2]) for f in range(1, F)]
        loop_loc = torch.cat([o[0] for o in loop], 0).clone()
        loop_conf = torch.cat([o[1].view(Bk, -1, 21) for o in loop], 0).clone()
        loop_offs = [t.clone() for t in first[2]]
        all_frames = frames.view(F * Bk, 3, 320, 320)
        got = temp(all_frames, ref_loc=maps, ret_off=True)
        assert torch.equal(got[0], loop_loc), (model, dtype)
        assert torch.equal(got[1].view(F * Bk, -1, 21), loop_conf), (model, dtype)
        assert len(got[2]) == 4 and all(torch.equal(a, b) for a, b in zip(got[2], loop_offs))
        # other frames of the same clips, offsets still in the workspace
        more = torch.from_numpy(synth.synth_frames(F * Bk, 320, seed=43)).to(DEV)
        again = temp(more, offset_list=got[2])
        want = torch.cat([temp(more[f * Bk:(f + 1) * Bk], ref_loc=maps)[0] for f in range(F)], 0)
        assert torch.equal(again[0], want), (model, dtype)
        # a key-frame count that does not divide the batch is an argument error
        with pytest.raises(ValueError):
            temp(all_frames[:F * Bk - 1], ref_loc=maps)
        if dtype == "fp32" and model == "ssd4scale_vgg":
            # ... and directly against the CPU oracle run the reference's way (evaluate_trn.py:452-462): key frame of clip 1 through
            # the static and the temporal net, frame 1 of that clip with the cached offset_list
            c = 1
            xk = frames[0, c:c + 1].cpu().numpy()
            _, _, r_maps = net_ref.ssd4scale_vgg_forward(sd_s, xk, 21, "test", True, False, ret_loc=True)
            r0_loc, r0_conf, r_offs = net_ref.ssd4scale_vgg_forward(sd_t, xk, 21, "test", True, True, ref_loc=r_maps, ret_off=True)
            r1_loc, r1_conf = net_ref.ssd4scale_vgg_forward(sd_t, frames[1, c:c + 1].cpu().numpy(), 21, "test", True, True, offset_list=r_offs)[:2]
            conf = got[1].view(F * Bk, -1, 21)
            np.testing.assert_allclose(got[0][c].cpu().numpy(), r0_loc.numpy()[0], atol=2e-3, rtol=0)
            np.testing.assert_allclose(got[0][Bk + c].cpu().numpy(), r1_loc.numpy()[0], atol=2e-3, rtol=0)
            np.testing.assert_allclose(conf[Bk + c].cpu().numpy(), r1_conf.numpy().reshape(-1, 21), atol=1e-3, rtol=0)


def test_trn_static_net_beside_the_temporal_trunk():
    """tdrn_net_io.reserved[2]: the static net's forward on a SECOND stream while the temporal net's trunk runs on the caller's; the
    temporal forward waits for the static net's event right before its first read of the loc maps.  Eager and as one captured
    hipGraph (the static net on a one-stream plan: a forward that forks its own lanes from a stream which joined the capture by an
    event takes hipStreamEndCapture down on ROCm 7.2): every output equals the serial order's, bit for bit, also when the static
    net is made slow enough that a missing wait would read stale maps (a second, different key-frame batch in the same buffers)."""
    stat, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, False), seed=0)
    stat.set_plan_flags(_lib.PLAN_ONE_STREAM)
    temp, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, True), seed=1)
    for n in (stat, temp):
        n.set_compute_dtype("bf16")
    Bk, F = 2, 3
    side, ev = torch.cuda.Stream(DEV), torch.cuda.Event()

    def serial(frames):
        s_loc, _, maps = stat(frames[0], ret_loc=True)
        loc, conf = temp(frames.view(F * Bk, 3, 320, 320), ref_loc=maps)[:2]
        return loc, conf, s_loc

    def overlapped(frames):
        main = torch.cuda.current_stream(DEV)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            s_loc, _, maps = stat(frames[0], ret_loc=True)
            ev.record(side)
        loc, conf = temp(frames.view(F * Bk, 3, 320, 320), ref_loc=maps, ref_event=ev)[:2]
        main.wait_stream(side)
        return loc, conf, s_loc

    batches = [torch.from_numpy(synth.synth_frames(F * Bk, 320, seed=51 + k)).to(DEV).view(F, Bk, 3, 320, 320) for k in range(3)]
    want = [[t.clone() for t in serial(b)] for b in batches]
    for b, w in zip(batches, want):                                   # eager
        got = overlapped(b)
        torch.cuda.synchronize()
        assert all(torch.equal(u, v) for u, v in zip(got, w))
    from tdrn_amd.engine import GraphedCall                           # one captured graph, replayed on changing inputs
    g = GraphedCall(overlapped, batches[0])
    for b, w in zip(batches, want):
        got = g(b)
        torch.cuda.synchronize()
        assert all(torch.equal(u, v) for u, v in zip(got, w))


def test_trn_two_clip_steps_in_flight_equal_one_at_a_time():
    """bench.py --config 5 with two steps in flight: the second pipeline is a `pipeline_twin()` of the static and of the temporal model
    (own engines -- workspace, lanes, offset state -- over the same weight blobs), its own side stream and event; the two pipelines are
    launched eagerly (InFlight(steps=...)), their streams picked by calibration.  Every step's outputs are the serial order's, bit for
    bit, turn after turn."""
    from tdrn_amd.engine import InFlight
    stat, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, False), seed=0)
    stat.set_plan_flags(_lib.PLAN_ONE_STREAM)
    temp, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, True), seed=1)
    for n in (stat, temp):
        n.set_compute_dtype("bf16")
    Bk, F, NB = 2, 3, 4
    batches = [torch.from_numpy(synth.synth_frames(F * Bk, 320, seed=71 + k)).to(DEV).view(F, Bk, 3, 320, 320) for k in range(NB)]

    def serial(frames):
        s_loc, _, maps = stat(frames[0], ret_loc=True)
        loc, conf = temp(frames.view(F * Bk, 3, 320, 320), ref_loc=maps)[:2]
        return loc, conf, s_loc
    want = [[t.clone() for t in serial(b)] for b in batches]
    torch.cuda.synchronize()

    def make(st, tp):
        side, ev = torch.cuda.Stream(DEV), torch.cuda.Event()

        def overlapped(frames):
            main = torch.cuda.current_stream(DEV)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                s_loc, _, maps = st(frames[0], ret_loc=True)
                ev.record(side)
            loc, conf = tp(frames.view(F * Bk, 3, 320, 320), ref_loc=maps, ref_event=ev)[:2]
            main.wait_stream(side)
            for t_ in [s_loc] + list(maps):
                t_.record_stream(main)
            return loc, conf, s_loc
        return overlapped
    stat2, temp2 = stat.pipeline_twin(DEV), temp.pipeline_twin(DEV)
    assert stat2.engine(DEV) is not stat.engine(DEV) and stat2.engine(DEV).weights.data_ptr() == stat.engine(DEV).weights.data_ptr()
    fl = InFlight(None, temp.engine(DEV), batches, graph=False, steps=[make(stat, temp), make(stat2, temp2)],
                  engines=[stat.engine(DEV), temp.engine(DEV), stat2.engine(DEV), temp2.engine(DEV)])
    assert fl.n == 2
    for phase in range(2):
        for turn in range(3):
            for k in range(NB):
                fl.launch(turn * NB + k)
            fl.sync()
            for j in range(NB):
                assert all(torch.equal(u, v) for u, v in zip(fl.output(j), want[j])), (phase, turn, j)
        if phase == 0:
            assert fl.pick_streams(candidates=3, steps=NB)["picked"]
    fl.check()


def test_pipeline_twin_follows_a_reload_of_its_source():
    """Round-5 advisor finding: a `pipeline_twin()` kept the blob of the moment it was made -- after load_state_dict /
    set_compute_dtype / set_plan_flags on the source the two pipelines ran different weights without any error.  The twin now
    follows its source: it re-clones whenever the source has re-packed (tdrn_amd/model/_base.py _twin_engine)."""
    net, _ = _build("ssd4scale_vgg", (320, 21, 1024, True, False), seed=0)
    net.set_compute_dtype("bf16")
    x = torch.from_numpy(synth.synth_frames(2, 320, seed=5)).to(DEV)
    twin = net.pipeline_twin(DEV)
    a, b = [t.clone() for t in net(x)[:2]], [t.clone() for t in twin(x)[:2]]
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    e_old = twin.engine(DEV)
    # new weights on the SOURCE only
    sd2 = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    c, d = [t.clone() for t in net(x)[:2]], [t.clone() for t in twin(x)[:2]]
    assert not torch.equal(a[0], c[0])                                     # the reload changed the result ...
    assert all(torch.equal(u, v) for u, v in zip(c, d))                    # ... on both pipelines
    assert twin.engine(DEV) is not e_old and twin.engine(DEV).weights is net.engine(DEV).weights
    # a dtype switch and a plan switch on the source reach the twin too; the twin used FIRST re-packs the source
    net.set_compute_dtype("fp16")
    d16 = [t.clone() for t in twin(x)[:2]]
    c16 = [t.clone() for t in net(x)[:2]]
    assert all(torch.equal(u, v) for u, v in zip(c16, d16)) and not torch.equal(c16[0], c[0])
    net.set_plan_flags(_lib.PLAN_ONE_STREAM)
    assert all(torch.equal(u, v) for u, v in zip(net(x)[:2], twin(x)[:2]))
    # a twin of a twin is a twin of the source
    assert net.pipeline_twin(DEV)._twin_of is net and twin.pipeline_twin(DEV)._twin_of is net


def test_two_steps_in_flight_equal_one_at_a_time():
    """Round 5, the default schedule of bench.py / FrameStream: two whole steps in flight (tdrn_amd.engine.InFlight -- pipeline p =
    its own engine handle, workspace, stream and hipGraph over ONE weight blob).  Every batch's detections are, bit for bit, what
    the same batch gives alone on one pipeline -- under concurrency (the chained split of conv3x3_pp polls flags while another
    step's kernels hold CUs), turn after turn, and tdrn_net_check stays clean."""
    from tdrn_amd.engine import InFlight
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
    net.set_compute_dtype("bf16")
    eng = net.engine(DEV)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    B, NB = 32, 4
    xb = [torch.from_numpy(synth.synth_frames(B, 320, seed=300 + j)).to(DEV) for j in range(NB)]

    def make_step(e):
        d = Detect(21, 0, 200, 0.01, 0.45)

        def one(x):
            r = e.forward(x)
            return d.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=[500.0, 375.0, 500.0, 375.0])
        return one
    alone = make_step(eng)
    want = []
    for j in range(NB):
        want.append(alone(xb[j]).clone())
        torch.cuda.synchronize()
    for graph in (True, False):
        fl = InFlight(make_step, eng, xb, n=2, graph=graph)
        assert len(fl.engines) == 2 and fl.engines[1].weights.data_ptr() == eng.weights.data_ptr()
        for turn in range(6):
            for k in range(NB):
                fl.launch(turn * NB + k)
            if turn % 2:
                fl.sync()
                for j in range(NB):
                    assert torch.equal(fl.output(j), want[j]), (graph, turn, j)
        fl.sync()
        fl.check()
        if not graph:
            # the eager pipelines' streams chosen by calibration (bench.py's default launch mode when it is the faster one): every
            # ordered pair of candidate streams is run; afterwards the results are still the same bits
            cal = fl.pick_streams(candidates=3, steps=NB)
            assert cal and cal["picked"] in cal["ms_per_step"] and len(cal["ms_per_step"]) == 6
            for turn in range(2):
                for k in range(NB):
                    fl.launch(turn * NB + k)
            fl.sync()
            for j in range(NB):
                assert torch.equal(fl.output(j), want[j]), ("picked streams", j)
            fl.check()
            # a consumer on ANOTHER stream that only waits for an event: the eager outputs are fresh allocations of the pipeline's stream,
            # so it asks for them with consumer=<its stream> (record_stream on every tensor; round-5 advisor finding) -- the next
            # launches of the same batch must not recycle the memory under it
            side = torch.cuda.Stream(DEV)
            for k in range(NB):
                fl.launch(k)
            copies = []
            for j in range(NB):
                ev = torch.cuda.Event()
                ev.record(fl.stream_of(j))
                side.wait_event(ev)
                out_j = fl.output(j, consumer=side)
                with torch.cuda.stream(side):
                    copies.append(out_j.clone())
            for k in range(2 * NB):                      # (re-launches that replace every output while `side` may still be copying)
                fl.launch(k)
            fl.sync()
            for j in range(NB):
                assert torch.equal(copies[j], want[j]), ("consumer stream", j)
        else:
            assert fl.pick_streams() is None             # (graph pipelines: the streams are part of the capture)
    assert (want[0][..., 0] > 0).any()


def test_detect_writes_into_a_given_buffer_device_or_pinned_host():
    """Detect(...).forward(..., out=): the rows land in the caller's buffer -- on the device, or in pinned host memory (the streamed mode's
    zero-copy hand-back) -- and are the rows the plain call returns; wrong shapes / pageable memory are refused."""
    rng = np.random.RandomState(5)
    B, P, Cn = 3, 6375, 21
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    loc = torch.from_numpy(rng.randn(B, P, 4).astype(np.float32) * 0.2).to(DEV)
    arm = torch.from_numpy(rng.randn(B, P, 4).astype(np.float32) * 0.2).to(DEV)
    conf = torch.softmax(torch.from_numpy(rng.randn(B * P, Cn).astype(np.float32) * 3), dim=1).to(DEV)
    det = Detect(Cn, 0, 200, 0.01, 0.45)
    want = det.forward(loc, conf, pri, arm_loc_data=arm).clone()
    assert (want[..., 0] > 0).any()
    on_dev = torch.full((B, Cn, 200, 5), -1.0, device=DEV)
    assert det.forward(loc, conf, pri, arm_loc_data=arm, out=on_dev) is on_dev and torch.equal(on_dev, want)
    pinned = torch.full((B, Cn, 200, 5), -1.0).pin_memory()
    assert det.forward(loc, conf, pri, arm_loc_data=arm, out=pinned) is pinned
    torch.cuda.synchronize()
    assert torch.equal(pinned, want.cpu())
    with pytest.raises(ValueError):
        det.forward(loc, conf, pri, arm_loc_data=arm, out=torch.empty((B, Cn, 200, 5)))              # pageable host memory
    with pytest.raises(ValueError):
        det.forward(loc, conf, pri, arm_loc_data=arm, out=torch.empty((B, Cn, 100, 5), device=DEV))


@pytest.mark.parametrize("graph,zero_copy,copy_in", [(True, False, "stream"), (False, False, "stream"), (False, True, "stream"), (True, True, "stream"),
                                                     (False, False, "own"), (True, False, "own")])
def test_frame_stream_with_two_pipelines_equals_unstreamed(graph, zero_copy, copy_in):
    """FrameStream given two engines (slot s on pipeline s % 2): same results as the unstreamed step, slot after slot -- as one hipGraph
    per slot, and with the steps launched eagerly on pipeline streams picked by calibration (the streamed twin of InFlight.pick_streams)."""
    from tdrn_amd.stream import FrameStream
    net, _ = _build("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))
    net.set_compute_dtype("bf16")
    eng = net.engine(DEV)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    B, slots, n = 4, 4, 11
    fs = FrameStream([eng, eng.clone()], Detect(21, 0, 200, 0.01, 0.45), pri, B, slots=slots, calibrate=not graph, graph=graph, zero_copy_out=zero_copy, copy_in=copy_in)
    assert fs.pipelines == 2
    if not graph:
        assert fs.pipeline_calibration["picked"] in fs.pipeline_calibration["ms_per_step"]
        assert (fs.calibration is None) == (copy_in == "own")          # (own: no copy streams to place)
    rng = np.random.RandomState(12)
    feeds = [torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3), dtype=np.uint8)) for _ in range(n)]
    fs.prime(feeds[:2])
    results, pending = [], []
    for k in range(n):
        if k + 2 < n:
            fs.pinned_in(fs.next_in()).copy_(feeds[k + 2])
        pending.append(fs.run())
        if len(pending) == 3:
            results.append(fs.result(pending.pop(0)).clone())
    while pending:
        results.append(fs.result(pending.pop(0)).clone())
    fs.drain()
    for k, got in enumerate(results):
        assert torch.equal(got, fs.eager(feeds[k].to(DEV)).cpu()), k
    if graph:
        with pytest.raises(ValueError):
            FrameStream([eng, eng.clone()], Detect(21, 0, 200, 0.01, 0.45), pri, B, slots=3, calibrate=False)


@pytest.mark.parametrize("model,args,dtype,batches", [
    ("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True), "bf16", (32, 1, 5)),      # conv3x3_ws.hip's producers read the uint8 planes
    ("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True), "fp16", (3,)),
    ("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True), "fp32", (2,)),             # first conv a launch of its own: planes converted first
    ("dualrefinedet_mobilenet", (320, 21, 1, True), "fp16", (4,)),                     # stride-2 first conv
    ("ssd4scale_vgg", (320, 21, 1024, False, False), "bf16", (8,)),
])
def test_uint8_frames_equal_fp32_frames(model, args, dtype, batches):
    """SURVEY 8f rank 1 in full: frames stay uint8 until the first conv reads them (tdrn_preprocess_u8 -> tdrn_net_io.reserved[3]).
    net(U8Frames) == net(base_transform(...)) bit for bit on every output, for the plans that read the planes inside conv1_2's producers
    and for the ones that convert them first; and the uint8 resize is the fp32 preprocess before its mean subtraction."""
    from tdrn_amd.data import base_transform, base_transform_u8
    net, _ = _build(model, args)
    net.set_compute_dtype(dtype)
    rng = np.random.RandomState(17)
    for B in batches:
        frames = torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3), dtype=np.uint8)).to(DEV)
        for to_rgb in (True, False):
            xf = base_transform(frames, 320, (104, 117, 123), to_rgb)
            xu = base_transform_u8(frames, 320, (104, 117, 123), to_rgb)
            assert xu.planes.dtype == torch.uint8 and torch.equal(xu.float(), xf)
        want = net(xf)
        got = net(xu)
        for u, v in zip(want, got):
            if torch.is_tensor(u):
                assert torch.equal(u, v), (model, dtype, B)
            elif u is not None:
                for uu, vv in zip(u, v):
                    assert torch.equal(uu, vv), (model, dtype, B)
