"""Pins the CPU oracle (oracle/) against the reference's own outputs (tests/golden/*.npz, made by
tests/golden/make_golden.py from the reference's Python) and against analytic known-answer tests
for the deformable op, whose reference implementation is CUDA-only."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net_ref
from oracle import oracle as orc
from tdrn_amd.utils import synth

VOC_320 = dict(feature_maps=[40, 20, 10, 5], min_dim=320, steps=[8, 16, 32, 64],
               min_sizes=[32, 64, 128, 256], max_sizes=[], aspect_ratios=[[2], [2], [2], [2]],
               variance=[0.1, 0.2], clip=True, flip=True, name="VOC_320")
VOC_512 = dict(VOC_320, feature_maps=[64, 32, 16, 8], min_dim=512, name="VOC_512_RefineDet")


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("cfg,name", [(VOC_320, "VOC_320"), (VOC_512, "VOC_512_RefineDet")])
def test_priorbox_bit_exact(golden_dir, cfg, name):
    ref = _g(golden_dir, "priorbox_%s.npz" % name)["priors"]
    got = orc.prior_box(cfg)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # closed-form first rows
    np.testing.assert_allclose(orc.prior_box(VOC_320)[:4], [[.0125, .0125, .1, .1],
                               [.0125, .0125, .14142136, .07071068],
                               [.0125, .0125, .07071068, .14142136], [.0375, .0125, .1, .1]], rtol=1e-6)


def test_decode_center_size_l2norm(golden_dir):
    g = _g(golden_dir, "box_utils.npz")
    pri = orc.prior_box(VOC_320)
    dec = orc.decode(g["loc"], pri)
    np.testing.assert_allclose(dec, g["decoded"], rtol=2e-6, atol=1e-7)   # expf vs SLEEF: 1 ulp
    # everything except the exp is exact: feed the reference's decoded boxes through center_size
    assert np.array_equal(orc.center_size(g["decoded"]), g["center_size"])
    np.testing.assert_allclose(orc.l2norm(g["l2_x"], g["l2_w"]), g["l2_y"], rtol=1e-6, atol=1e-7)


def test_nms_matches_reference_numpy_twin(golden_dir):
    g = _g(golden_dir, "nms_cases.npz")
    n_cases = len([k for k in g.files if k.startswith("dets")])
    assert n_cases >= 7
    for i in range(n_cases):
        dets, keep = g["dets%d" % i], g["keep%d" % i]
        assert len(np.unique(dets[:, 4])) == len(dets), "fixture must be tie-free"
        got = np.asarray(orc.cpu_nms(dets, 0.45), np.int32)
        assert np.array_equal(got, keep), "case %d" % i


def test_nms_threshold_equality_rule():
    # two boxes with IoU exactly 0.5 (+1 convention): 10x10 and 10x10 shifted so inter=... ;
    # a = [0,0,9,9] (area 100), b = [0,0,9,4]+... choose inter/union = 50/100
    a = [0, 0, 9, 9, 0.9]
    b = [0, 0, 9, 4, 0.8]                  # area 50, inter 50, union 100 -> ovr = 0.5 exactly
    dets = np.asarray([a, b], np.float32)
    assert orc.cpu_nms(dets, 0.5) == [0]                   # cpu_nms.pyx:66  ovr >= thresh
    assert orc.cpu_nms(dets, 0.5, strict_gt=True) == [0, 1]  # nms_kernel.cu:71 ovr > thresh
    assert orc.cpu_nms(dets, 0.5000001) == [0, 1]
    assert orc.cpu_nms(np.zeros((0, 5), np.float32), 0.5) == []


@pytest.mark.parametrize("tag", ["D8", "D9", "D6"])
def test_detect_matches_reference(golden_dir, tag):
    g = _g(golden_dir, "detect_%s.npz" % tag)
    B = int(g["batch"])
    loc, arm, conf = synth.synth_detect_inputs(B, 6375, 21, float(g["bias"]), seed=1)
    pri = orc.prior_box(VOC_320)
    out = orc.detect(loc, conf, pri, arm, (500, 375, 500, 375))
    ref = g["out"]
    assert np.array_equal(out[..., 0], ref[..., 0])        # scores/slot occupancy: exact
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)
    out2 = orc.detect(loc, conf, pri, None, (320,) * 4)
    assert np.array_equal(out2[..., 0], g["out_noarm"][..., 0])
    np.testing.assert_allclose(out2, g["out_noarm"], rtol=3e-6, atol=1e-6)


# ---- deformable conv known-answer tests (utils/deformconv/deform_conv_cuda_kernel.cu:15-51,156-208)
def _rand(shape, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("k,pad", [(3, 1), (5, 2), (1, 0)])
def test_deform_zero_offset_is_plain_conv(k, pad):
    x, w = _rand((2, 6, 9, 7), 1), _rand((5, 6, k, k), 2)
    off = np.zeros((2, 2 * k * k, 9, 7), np.float32)
    got = orc.deform_conv_forward(x, off, w, 1, pad, 1, 1)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, 1, pad).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_deform_integer_offset_is_shifted_conv():
    # constant offset (+1 row, +2 cols): sample x[h+1, w+2]; rows/cols past the far edge clamp...
    x, w = _rand((1, 4, 8, 8), 3), _rand((3, 4, 3, 3), 4)
    off = np.zeros((1, 18, 8, 8), np.float32)
    off[:, 0::2] = 1.0
    off[:, 1::2] = 2.0
    got = orc.deform_conv_forward(x, off, w, 1, 1, 1, 1)
    # build the shifted image with the kernel's border rule: coordinates >= H (or W) -> 0,
    # negative -> 0, integer coords inside -> exact value.
    xs = np.zeros((1, 4, 8 + 2 + 4, 8 + 2 + 4), np.float32)      # generous zero canvas
    xs[:, :, 2:10, 2:10] = x
    # out(h,w) tap (i,j) reads x[h-1+i+1, w-1+j+2] = canvas[h+i+2, w+j+3]
    shifted = xs[:, :, 2:, 3:]
    ref = F.conv2d(torch.from_numpy(np.ascontiguousarray(shifted)), torch.from_numpy(w)).numpy()
    np.testing.assert_allclose(got, ref[:, :, :8, :8], rtol=1e-4, atol=1e-5)


def test_deform_border_rules_single_pixel():
    H = W = 4
    x = np.arange(16, dtype=np.float32).reshape(1, 1, H, W) + 1.0
    w = np.ones((1, 1, 1, 1), np.float32)

    def sample(h, wq, dh, dw):
        off = np.zeros((1, 2, H, W), np.float32)
        off[0, 0, h, wq], off[0, 1, h, wq] = dh, dw
        return float(orc.deform_conv_forward(x, off, w, 1, 0, 1, 1)[0, 0, h, wq])
    assert sample(1, 1, 0.5, 0.0) == pytest.approx((x[0, 0, 1, 1] + x[0, 0, 2, 1]) / 2)
    assert sample(0, 0, -0.25, 0.0) == 0.0                     # h_im in (-1,0): hard zero (:195)
    assert sample(0, 0, 0.0, -1e-3) == 0.0
    assert sample(3, 2, 0.75, 0.0) == x[0, 0, 3, 2]             # h_im in (H-1,H): clamps to row H-1
    assert sample(2, 3, 0.0, 0.5) == x[0, 0, 2, 3]              # w_im in (W-1,W): clamps to col W-1
    assert sample(3, 3, 1.0, 0.0) == 0.0                        # h_im == H: rejected
    assert sample(3, 3, 0.999, 0.999) == x[0, 0, 3, 3]
    v = sample(2, 2, 0.5, 0.5)                                 # interior bilinear
    assert v == pytest.approx(x[0, 0, 2:4, 2:4].mean())


def test_deform_group_indexing():
    # G=2: channels [0,2) use offset group 0, [2,4) group 1 (.cu:172)
    x, w = _rand((1, 4, 6, 6), 5), _rand((2, 4, 3, 3), 6)
    off = np.zeros((1, 2 * 18, 6, 6), np.float32)
    off[:, 18:] = _rand((1, 18, 6, 6), 7) * 0.7
    got = orc.deform_conv_forward(x, off, w, 1, 1, 1, 2)
    x0, x1 = x.copy(), x.copy()
    x0[:, 2:] = 0
    x1[:, :2] = 0
    a = orc.deform_conv_forward(x0, off[:, :18], w, 1, 1, 1, 1)
    b = orc.deform_conv_forward(x1, off[:, 18:], w, 1, 1, 1, 1)
    np.testing.assert_allclose(got, a + b, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        orc.deform_conv_forward(x, off[:, :18], w, 1, 1, 1, 2)  # offset channels != G*2*kh*kw


@pytest.mark.parametrize("tag,mh", [("drn_vggbn_320_mh", True), ("drn_vggbn_320", False)])
def test_full_net_restatement_matches_reference(golden_dir, tag, mh):
    """oracle/net_ref.py vs the reference's RefineSSD.forward (deformable op patched to the oracle)."""
    g = _g(golden_dir, tag + ".npz")
    keys = [str(k) for k in g["keys"]]
    shapes = _drn_vggbn_shapes(mh)
    assert sorted(shapes.keys()) == sorted(keys)          # same state_dict keys as the reference
    sd = synth.synth_state_dict(shapes, 0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    arm, offs, odm, conf = net_ref.drn_vggbn_forward(sd, synth.synth_frames(1, 320, 0), 21, True, mh)
    sub = int(g["sub"])
    P = arm.shape[1]
    np.testing.assert_allclose(arm.numpy()[:, ::sub], g["arm_loc"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(odm.numpy()[:, ::sub], g["odm_loc"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(conf.numpy().reshape(1, P, 21)[:, ::sub], g["conf"], rtol=1e-4,
                               atol=2e-5)
    np.testing.assert_allclose(offs[3].numpy(), g["off3"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(offs[0].numpy()[:, :, ::5, ::5], g["off0"], rtol=1e-4, atol=2e-5)
    assert float(g["stats"][3]) == pytest.approx(float(arm.double().sum()), rel=1e-4, abs=1e-2)


def _drn_vggbn_shapes(multihead, c7=1024, nc=21):
    """State-dict layout of model/dualrefinedet_vggbn.py, incl. num_batches_tracked."""
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    net = build_net("test", 320, nc, c7, 1, True, multihead)
    return {k: tuple(v.shape) for k, v in net.state_dict().items()}


def _shapes(modname, args):
    import importlib
    net = importlib.import_module("tdrn_amd.model." + modname).build_net("test", *args)
    return {k: tuple(v.shape) for k, v in net.state_dict().items()}


@pytest.mark.parametrize("tag,mh", [("mh", True), ("sh", False)])
def test_drn_mobilenet_restatement_matches_reference(golden_dir, tag, mh):
    """BASELINE config #4's model: oracle/net_ref.drn_mobilenet_forward vs the reference's own
    model/dualrefinedet_mobilenet.py:127-199 forward (deformable op patched to the oracle), multihead on / off."""
    g = _g(golden_dir, "drn_mobilenet_320.npz")
    sub = int(g["sub"])
    sh = _shapes("dualrefinedet_mobilenet", (320, 21, 1, mh))
    assert sorted(sh) == sorted(str(k) for k in g[tag + "_keys"])
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    x = synth.synth_frames(1, 320, int(g["x_seed"]))
    arm, none, odm, conf = net_ref.drn_mobilenet_forward(synth.synth_state_dict(sh, 0), x, 21, mh)
    assert none is None
    np.testing.assert_allclose(arm.numpy()[:, ::sub], g[tag + "_arm"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(odm.numpy()[:, ::sub], g[tag + "_odm"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conf.numpy().reshape(1, -1, 21)[:, ::sub], g[tag + "_conf"], rtol=1e-4, atol=5e-5)


def test_other_model_restatements_match_reference(golden_dir):
    """refinedet_vgg (the reference's own CPU forward, no patched op at all), ssd4scale_vgg / _mobile static
    and temporal (TRN) nets: oracle/net_ref.py vs tests/golden/other_models.npz."""
    g = _g(golden_dir, "other_models.npz")
    sub = int(g["sub"])
    x = synth.synth_frames(1, 320, 21)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    tol = dict(rtol=1e-4, atol=5e-5)

    sh = _shapes("refinedet_vgg", (320, 21, True, 1024, True, True))
    assert sorted(sh) == sorted(str(k) for k in g["rd_keys"])
    arm, _, odm, conf = net_ref.refinedet_vgg_forward(synth.synth_state_dict(sh, 0), x, 21, True, True, True)
    np.testing.assert_allclose(arm.numpy()[:, ::sub], g["rd_arm"], **tol)
    np.testing.assert_allclose(odm.numpy()[:, ::sub], g["rd_odm"], **tol)
    np.testing.assert_allclose(conf.numpy().reshape(1, -1, 21)[:, ::sub], g["rd_conf"], **tol)
    sh = _shapes("refinedet_vgg", (320, 21, False, 1024, False, False))
    assert sorted(sh) == sorted(str(k) for k in g["rd0_keys"])
    odm, conf = net_ref.refinedet_vgg_forward(synth.synth_state_dict(sh, 0), x, 21, False, False, False)
    np.testing.assert_allclose(odm.numpy()[:, ::sub], g["rd0_odm"], **tol)
    np.testing.assert_allclose(conf.numpy().reshape(1, -1, 21)[:, ::sub], g["rd0_conf"], **tol)

    for tag, modname, a_s, a_t, fwd in (
            ("sv", "ssd4scale_vgg", (320, 21, 1024, True, False), (320, 21, 1024, True, True),
             lambda sd, **kw: net_ref.ssd4scale_vgg_forward(sd, x, 21, "test", True, **kw)),
            ("sm", "ssd4scale_mobile", (320, 21, 1024, False), (320, 21, 1024, True),
             lambda sd, **kw: net_ref.ssd4scale_mobile_forward(sd, x, 21, "test", **kw))):
        ss, ts = _shapes(modname, a_s), _shapes(modname, a_t)
        assert sorted(ss) == sorted(str(k) for k in g[tag + "_keys"]) and sorted(ts) == sorted(str(k) for k in g[tag + "_tkeys"])
        loc, conf, maps = fwd(synth.synth_state_dict(ss, 0), deform_on=False, ret_loc=True)
        np.testing.assert_allclose(loc.numpy()[:, ::sub], g[tag + "_loc"], **tol)
        np.testing.assert_allclose(conf.numpy().reshape(1, -1, 21)[:, ::sub], g[tag + "_conf"], **tol)
        np.testing.assert_allclose(maps[3].numpy(), g[tag + "_map3"], **tol)
        tloc, tconf, offs = fwd(synth.synth_state_dict(ts, 1), deform_on=True, ref_loc=maps, ret_off=True)
        np.testing.assert_allclose(tloc.numpy()[:, ::sub], g[tag + "_tloc"], **tol)
        np.testing.assert_allclose(tconf.numpy().reshape(1, -1, 21)[:, ::sub], g[tag + "_tconf"], **tol)
        np.testing.assert_allclose(offs[3].numpy(), g[tag + "_off3"], **tol)


# ---------------------------------------------------------------------------------------------
# preprocess oracle (cv2 is absent from the image: no fixture from the reference is possible, so the restatement of
# cv2.resize(INTER_LINEAR, 8-bit) is pinned by known-answer cases whose answers do not come from its own code)
# ---------------------------------------------------------------------------------------------
def test_preprocess_identity_is_pixel_minus_mean():
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    mean = (104, 117, 123)
    out = orc.base_transform_u8(img, 64, mean, to_rgb=False)
    assert np.array_equal(out, (img.astype(np.float32) - np.asarray(mean, np.float32)).transpose(0, 3, 1, 2))
    rgb = orc.base_transform_u8(img, 64, mean, to_rgb=True)       # voc0712.py:467-468: channels reversed AFTER the mean
    assert np.array_equal(rgb, out[:, ::-1])


def test_preprocess_exact_two_to_one_is_the_rounded_block_mean():
    """dst = src/2: every sample sits exactly between two source pixels (coefficients 1024/1024 of 2048), so the
    result is the 2x2 block mean; cv2's fixed-point pipeline rounds it half up: (a+b+c+d+2) >> 2."""
    rng = np.random.Generator(np.random.PCG64(4))
    img = rng.integers(0, 256, (1, 32, 48, 3), dtype=np.uint8)
    out = orc.base_transform_u8(img, 16, (0, 0, 0))               # non-square source on purpose: 32x48 -> 16x16 is 2:1 / 3:1
    blocks = img[0, :, :, :].astype(np.int64).reshape(16, 2, 48, 3).sum(1)          # vertical pairs
    got_rows = out[0].transpose(1, 2, 0)                                            # (16,16,3)
    # horizontal 3:1: sample x = 3*dx + 1 exactly (fraction 0) -> column 3dx+1 alone; vertical 2:1 -> mean of the pair
    expect = (2 * blocks[:, 1::3, :] + 2) >> 2                                      # (a+a+c+c+2)>>2 with identical horizontal pair
    assert np.array_equal(got_rows, expect.astype(np.float32))
    sq = rng.integers(0, 256, (1, 32, 32, 3), dtype=np.uint8)
    out = orc.base_transform_u8(sq, 16, (0, 0, 0))[0].transpose(1, 2, 0)
    s = sq[0].astype(np.int64)
    expect = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(out, expect.astype(np.float32))


def test_preprocess_exact_one_to_two_quarter_weights_and_border_clamp():
    """dst = 2*src: samples fall at k -/+ 0.25 -> weights (0.25, 0.75) = (512, 1536)/2048; the outermost samples
    (source coordinate -0.25 and n-0.75) clamp to the border pixel (cv2: sx < 0 -> sx = 0, fx = 0; sx >= n-1 -> fx = 0).
    On a horizontal ramp with constant rows the vertical pass is the identity, so the answer is a 1-D formula."""
    ramp = (np.arange(8) * 32).astype(np.uint8)                 # 0, 32, ..., 224
    img = np.broadcast_to(ramp[None, None, :, None], (1, 8, 8, 3)).copy()
    out = orc.base_transform_u8(img, 16, (0, 0, 0))[0, 0]        # (16,16), all rows equal
    assert (out == out[0]).all()
    p = ramp.astype(np.int64)
    exp = np.empty(16, np.int64)
    exp[0] = p[0]
    exp[15] = p[7]
    for k in range(1, 8):
        exp[2 * k] = (512 * p[k - 1] + 1536 * p[k] + 1024) >> 11          # k - 0.25
    for k in range(0, 7):
        exp[2 * k + 1] = (1536 * p[k] + 512 * p[k + 1] + 1024) >> 11      # k + 0.25
    assert np.array_equal(out[0], exp.astype(np.float32))


def test_preprocess_hand_computed_3x3_to_2x2():
    """3x3 -> 2x2: source coordinates 0.25 and 1.75 in both axes -> coefficient pairs (1536, 512) and (512, 1536).
    Worked by hand with OpenCV's two passes (horizontal sums kept at 11 fractional bits, vertical pass
    (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2):
        src = [[ 10,  20,  30],      H-pass rows: [1536*10+512*20, 512*20+1536*30] = [25600, 56320]
               [ 40,  50,  60],                   [1536*40+512*50, 512*50+1536*60] = [87040, 117760]
               [ 70,  80,  90]]                   [1536*70+512*80, 512*80+1536*90] = [148480, 179200]
        dst[0][0] = ((1536*(25600>>4))>>16) + ((512*(87040>>4))>>16) + 2 >> 2 = (37 + 42 + 2) >> 2 = 20
        dst[0][1] = ((1536*3520)>>16) + ((512*7360)>>16) + 2 >> 2 = (82 + 57 + 2) >> 2 = 35
        dst[1][0] = ((512*5440)>>16) + ((1536*9280)>>16) + 2 >> 2 = (42 + 217 + 2) >> 2 = 65
        dst[1][1] = ((512*7360)>>16) + ((1536*11200)>>16) + 2 >> 2 = (57 + 262 + 2) >> 2 = 80"""
    src = np.asarray([[10, 20, 30], [40, 50, 60], [70, 80, 90]], np.uint8)
    img = np.repeat(src[None, :, :, None], 3, axis=3)
    out = orc.base_transform_u8(img, 2, (0, 0, 0))[0]
    assert np.array_equal(out[0], np.asarray([[20, 35], [65, 80]], np.float32))
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[1], out[2])


def test_preprocess_resize_within_one_lsb_of_an_independent_bilinear():
    """cv2 itself is absent, so the fixed-point pipeline cannot be compared with cv2's output; what CAN be pinned from outside this
    repo is the sampling geometry: torch's `F.interpolate(mode="bilinear", align_corners=False)` is an independent implementation of
    the same half-pixel-centre, edge-replicating bilinear rule in floating point, and an 11-bit fixed-point evaluation of that rule
    (coefficients rounded to 1/2048, two truncating shifts, one rounding) stays within ONE grey level of it -- on random uint8 frames,
    up- and down-scaling, square and not (the VOC frame shape 375 x 500 -> 320 / 512 included).  A wrong source coordinate, a wrong
    border rule or swapped coefficient pairs show as errors of tens of grey levels."""
    import torch
    import torch.nn.functional as F
    rng = np.random.Generator(np.random.PCG64(9))
    for (h0, w0, s) in [(375, 500, 320), (375, 500, 512), (64, 48, 96), (33, 57, 40), (120, 120, 64), (7, 5, 16)]:
        img = rng.integers(0, 256, (2, h0, w0, 3), dtype=np.uint8)
        got = orc.base_transform_u8(img, s, (0, 0, 0))                         # (B,3,S,S), BGR order kept, mean 0
        ref = F.interpolate(torch.from_numpy(img).permute(0, 3, 1, 2).double(), size=(s, s), mode="bilinear", align_corners=False).numpy()
        d = np.abs(got.astype(np.float64) - ref)
        assert d.max() <= 1.0 + 1e-9, (h0, w0, s, float(d.max()))
        # ... and mostly it IS the rounded float result (the two truncating shifts of the vertical pass bias it down: 85-100 %)
        assert np.mean(np.abs(got - np.rint(ref)) == 0) > 0.8, (h0, w0, s)
        assert np.mean(got.astype(np.float64) - ref) < 0.0 + 0.05
