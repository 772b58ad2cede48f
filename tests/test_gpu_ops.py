"""GPU parity tests (run with -m gpu on an MI355X): every kernel of the path, through the C ABI /
its Python mirror, against the CPU oracle on the same seeded inputs, plus the golden fixtures
made from the reference.  Bit-exact for indices / keep lists; fp32 path within 1e-3 (the
tolerance BASELINE.json's north_star states); bf16/fp16 drift bounded separately."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tdrn_amd import _lib
from tdrn_amd.data import mb_cfg
from tdrn_amd.layers import Detect, PriorBox
from tdrn_amd.layers.box_utils import center_size, decode
from tdrn_amd.model.networks import ConvOffset2d, conv_offset2d
from tdrn_amd.utils import synth
from tdrn_amd.utils.nms_wrapper import cpu_nms, gpu_nms, nms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, seed, scale=1.0):
    return (scale * np.random.Generator(np.random.PCG64(seed)).standard_normal(shape)).astype(np.float32)


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ---------------------------------------------------------------------------------------------
# deformable conv (tdrn_deform_conv_forward) vs oracle (deform_conv_cuda_kernel.cu restatement)
# ---------------------------------------------------------------------------------------------
DEFORM_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil, G, offset scale
    (2, 6, 9, 7, 4, 3, 1, 1, 1, 1, 0.0),        # zero offsets == plain conv (reference test.py shape class)
    (1, 6, 9, 7, 4, 3, 1, 1, 1, 1, 1.0),
    (2, 32, 10, 10, 12, 3, 1, 1, 1, 1, 1.5),
    (1, 64, 20, 20, 75, 3, 1, 1, 1, 1, 1.0),    # fused loc+conf width
    (1, 64, 12, 11, 63, 5, 1, 2, 1, 1, 2.0),    # 5x5 multihead branch
    (2, 64, 8, 8, 12, 3, 1, 1, 1, 8, 1.0),      # 8 deformable groups (TRN heads)
    (1, 24, 13, 9, 10, 3, 2, 1, 1, 2, 1.0),     # stride 2, G=2, ragged channels
    (1, 16, 9, 9, 8, 3, 1, 2, 2, 1, 1.0),       # dilation 2
    (1, 8, 6, 6, 140, 1, 1, 0, 1, 1, 0.7),      # Cout > 128 (two channel chunks), 1x1
    (3, 256, 5, 5, 75, 3, 1, 1, 1, 1, 3.0),     # smallest pyramid level, big offsets (all borders)
    # non-square kernel / stride / padding / dilation (shape_check, deform_conv_cuda.c:7-96, takes them per axis)
    (2, 16, 11, 13, 9, (3, 5), 1, (1, 2), 1, 1, 1.0),
    (1, 32, 14, 9, 12, (1, 3), (2, 1), (0, 1), 1, 2, 1.5),
    (1, 8, 12, 12, 6, (3, 2), (1, 2), (2, 0), (2, 1), 1, 1.0),
]
_pr = lambda v: (v, v) if isinstance(v, int) else tuple(v)


@pytest.mark.parametrize("case", DEFORM_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_deform_conv_fp32_matches_oracle(case):
    N, Cin, H, W, Cout, k, st, pad, dil, G, osc = case
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = _pr(k), _pr(st), _pr(pad), _pr(dil)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    x, w = _rand((N, Cin, H, W), 1), _rand((Cout, Cin, kh, kw), 2, (Cin * kh * kw) ** -0.5)
    off = _rand((N, G * 2 * kh * kw, Ho, Wo), 3, osc)
    ref = orc.deform_conv_forward(x, off, w, st, pad, dil, G)
    got = conv_offset2d(_cu(x), _cu(off), _cu(w), st, pad, dil, G).cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("compute,tol", [("bf16", 3e-2), ("fp16", 4e-3)])
def test_deform_conv_16bit_drift(compute, tol):
    N, Cin, H, W, Cout, k = 2, 256, 10, 10, 75, 3
    x, w, off = _rand((N, Cin, H, W), 4), _rand((Cout, Cin, k, k), 5, (Cin * 9) ** -0.5), _rand((N, 18, H, W), 6)
    ref = orc.deform_conv_forward(x, off, w, 1, 1, 1, 1)
    got = conv_offset2d(_cu(x), _cu(off), _cu(w), 1, 1, 1, 1, compute=compute).cpu().numpy()
    assert np.abs(got - ref).max() < tol * max(1.0, np.abs(ref).max())


def test_deform_border_rules_on_device():
    H = W = 4
    x = np.arange(16, dtype=np.float32).reshape(1, 1, H, W) + 1.0
    w = np.ones((1, 1, 1, 1), np.float32)
    probes = [(1, 1, 0.5, 0.0), (0, 0, -0.25, 0.0), (0, 0, 0.0, -1e-3), (3, 2, 0.75, 0.0), (2, 3, 0.0, 0.5),
              (3, 3, 1.0, 0.0), (3, 3, 0.999, 0.999), (2, 2, 0.5, 0.5)]
    for h, wq, dh, dw in probes:
        off = np.zeros((1, 2, H, W), np.float32)
        off[0, 0, h, wq], off[0, 1, h, wq] = dh, dw
        ref = orc.deform_conv_forward(x, off, w, 1, 0, 1, 1)
        got = conv_offset2d(_cu(x), _cu(off), _cu(w), 1, 0, 1, 1).cpu().numpy()
        assert np.array_equal(got, ref), (h, wq, dh, dw, got[0, 0, h, wq], ref[0, 0, h, wq])


def test_convoffset2d_module_and_shape_errors():
    m = ConvOffset2d(6, 4, 3, padding=1).to(DEV)
    x, off = _cu(_rand((2, 6, 5, 5), 7)), _cu(_rand((2, 18, 5, 5), 8))
    y = m(x, off)
    ref = orc.deform_conv_forward(x.cpu().numpy(), off.cpu().numpy(), m.weight.detach().cpu().numpy(), 1, 1, 1, 1)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
    with pytest.raises(RuntimeError):
        m(x, off[:, :16])                                   # offset channels != G*2*kh*kw
    with pytest.raises(RuntimeError):
        m(x, off[:1])                                       # offset batch != input batch (.c:137)
    with pytest.raises(RuntimeError):
        m(x[:, :4], off)                                    # input planes


# ---------------------------------------------------------------------------------------------
# NMS / decode / Detect
# ---------------------------------------------------------------------------------------------
def _random_dets(n, spread, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    xy = rng.uniform(0, spread, (n, 2)).astype(np.float32)
    wh = rng.uniform(4, 60, (n, 2)).astype(np.float32)
    sc = rng.permutation(n).astype(np.float32) / np.float32(n) * 0.98 + 0.01
    return np.concatenate([xy, xy + wh, sc[:, None]], 1).astype(np.float32)


def test_nms_golden_cases_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "nms_cases.npz"))
    for i in range(len([k for k in g.files if k.startswith("dets")])):
        dets, keep = g["dets%d" % i], g["keep%d" % i]
        assert cpu_nms(dets, 0.45) == keep.tolist(), "case %d" % i
        assert nms(dets, 0.45, force_cpu=True) == keep.tolist()


@pytest.mark.parametrize("n,spread", [(1, 10), (2, 3), (63, 30), (64, 30), (65, 30), (129, 50), (777, 100),
                                      (4096, 400), (6375, 300), (16320, 500)])
def test_nms_random_matches_oracle(n, spread):
    dets = _random_dets(n, spread, 100 + n)
    for thr in (0.45, 0.3, 0.7):
        assert cpu_nms(dets, thr) == orc.cpu_nms(dets, thr)
    got = [int(v) for v in gpu_nms(dets, 0.45)]
    assert got == orc.cpu_nms(dets, 0.45, strict_gt=True)     # tie-free data: same order either way


def test_nms_threshold_equality_and_empty():
    dets = np.asarray([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], np.float32)   # IoU exactly 0.5
    assert cpu_nms(dets, 0.5) == [0]
    assert [int(v) for v in gpu_nms(dets, 0.5)] == [0, 1]
    assert cpu_nms(dets, 0.5000001) == [0, 1]
    assert nms(np.zeros((0, 5), np.float32), 0.5) == []
    # thresh is a double in cpu_nms.pyx: 0.45 (double) vs fp32 IoU just above/below it
    a = np.asarray([[0, 0, 99, 99, 0.9], [0, 0, 99, 44, 0.8]], np.float32)  # IoU = 0.45 in fp32
    assert cpu_nms(a, 0.45) == orc.cpu_nms(a, 0.45)


def test_decode_center_size(golden_dir):
    g = np.load(os.path.join(golden_dir, "box_utils.npz"))
    pri = PriorBox(mb_cfg["VOC_320"]).forward()
    dec = decode(_cu(g["loc"]), pri, [0.1, 0.2]).cpu().numpy()
    np.testing.assert_allclose(dec, g["decoded"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(dec, orc.decode(g["loc"], pri.numpy()), rtol=3e-6, atol=1e-7)
    assert np.array_equal(center_size(_cu(g["decoded"])).cpu().numpy(), g["center_size"])


@pytest.mark.parametrize("tag", ["D8", "D9", "D6"])
def test_detect_matches_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "detect_%s.npz" % tag))
    B = int(g["batch"])
    loc, arm, conf = synth.synth_detect_inputs(B, 6375, 21, float(g["bias"]), seed=1)
    pri = PriorBox(mb_cfg["VOC_320"]).forward()
    det = Detect(21, 0, 200, 0.01, 0.45)
    out = det.forward(_cu(loc), _cu(conf), pri.to(DEV), arm_loc_data=_cu(arm),
                      scale=torch.tensor([500.0, 375.0, 500.0, 375.0])).cpu().numpy()
    ref = g["out"]
    assert out.shape == ref.shape == (B, 21, 200, 5)
    assert np.array_equal(out[..., 0], ref[..., 0])          # scores / slot occupancy: exact
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)
    assert not out[:, 0].any()                                # background row stays zero
    out2 = det.forward(_cu(loc), _cu(conf), pri.to(DEV), feature=None).cpu().numpy()   # default scale, no ARM
    assert np.array_equal(out2[..., 0], g["out_noarm"][..., 0])
    np.testing.assert_allclose(out2, g["out_noarm"], rtol=3e-6, atol=1e-6)
    cnt = det.last_counts.cpu().numpy()
    assert np.array_equal(cnt, (out2[..., 0] > 0).sum(-1))


def test_detect_full_batch_properties_and_worst_case():
    """BASELINE sizes (B=32): per-image independence, writable output, worst case (every prior of
    every class is a candidate) against the oracle on one image."""
    B, P = 32, 6375
    loc, arm, conf = synth.synth_detect_inputs(B, P, 21, 8.0, seed=3)
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.01, 0.45)
    sc = [500.0, 375.0, 500.0, 375.0]
    full = det.forward(_cu(loc), _cu(conf), pri, arm_loc_data=_cu(arm), scale=sc)
    for b in (0, 13, 31):
        one = det.forward(_cu(loc[b:b + 1]), _cu(conf[b * P:(b + 1) * P]), pri, arm_loc_data=_cu(arm[b:b + 1]), scale=sc)
        assert torch.equal(one[0], full[b])
    ref = orc.detect(loc[5:6], conf[5 * P:6 * P], pri.cpu().numpy(), arm[5:6], sc)
    assert np.array_equal(full[5].cpu().numpy()[..., 0], ref[0][..., 0])
    full[0, 1, :, 1] *= 500.0                                  # callers scale boxes in place (evaluate.py:476-479)
    # worst case W: uniform conf -> all priors pass for every class
    confw = np.full((P, 21), 1.0 / 21, np.float32) + _rand((P, 21), 9, 1e-4)
    out = det.forward(_cu(loc[:1]), _cu(confw), pri, arm_loc_data=_cu(arm[:1]), scale=sc).cpu().numpy()
    refw = orc.detect(loc[:1], confw, pri.cpu().numpy(), arm[:1], sc)
    assert np.array_equal(out[..., 0], refw[..., 0])
    np.testing.assert_allclose(out, refw, rtol=3e-6, atol=1e-6)


def test_detect_selection_fallbacks_match_oracle():
    """More than 2048 candidates per class, arranged so that the top-2048 prefix cannot finish the job:
    (a) every box identical -> one survivor, the prefix runs out with candidates left (second launch);
    (b) all scores equal -> no strict score prefix exists (second launch, ties by ascending prior index);
    (c) scores in two tight clusters -> the radix select has to descend to the low bytes;
    (d) P = 16320 (the 512x512 prior count) with every prior a candidate."""
    P = 6375
    pri = PriorBox(mb_cfg["VOC_320"]).forward()
    det = Detect(21, 0, 200, 0.01, 0.45)
    sc = [500.0, 375.0, 500.0, 375.0]
    rng = np.random.Generator(np.random.PCG64(77))
    prin = pri.numpy()
    # (a) loc that maps every prior onto the same box (cx=cy=.5, w=h=.3): invert decode()
    loc = np.empty((1, P, 4), np.float32)
    loc[0, :, 0] = (0.5 - prin[:, 0]) / (0.1 * prin[:, 2])
    loc[0, :, 1] = (0.5 - prin[:, 1]) / (0.1 * prin[:, 3])
    loc[0, :, 2] = np.log(0.3 / prin[:, 2]) / 0.2
    loc[0, :, 3] = np.log(0.3 / prin[:, 3]) / 0.2
    conf = (0.02 + 0.9 * rng.random((P, 21))).astype(np.float32)
    out = det.forward(_cu(loc), _cu(conf), pri.to(DEV), scale=sc).cpu().numpy()
    ref = orc.detect(loc, conf, prin, None, sc)
    assert np.array_equal(out[..., 0], ref[..., 0])
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)
    assert (det.last_counts.cpu().numpy()[0, 1:] <= 3).all()
    # (b) + (c) on spread-out boxes
    loc2, arm2, _ = synth.synth_detect_inputs(1, P, 21, 8.0, seed=5)
    confb = np.full((P, 21), 0.04, np.float32)
    confc = np.where(rng.random((P, 21)) < 0.5, 0.04, 0.0400001).astype(np.float32) + (rng.integers(0, 64, (P, 21)) * 2.0 ** -30).astype(np.float32)
    for cf in (confb, confc):
        out = det.forward(_cu(loc2), _cu(cf), pri.to(DEV), arm_loc_data=_cu(arm2), scale=sc).cpu().numpy()
        ref = orc.detect(loc2, cf, prin, arm2, sc)
        assert np.array_equal(out[..., 0], ref[..., 0])
        np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)
    # (d)
    pri5 = PriorBox(mb_cfg["VOC_512_RefineDet"]).forward()
    P5 = pri5.shape[0]
    loc5, arm5, _ = synth.synth_detect_inputs(1, P5, 21, 8.0, seed=6)
    conf5 = (np.full((P5, 21), 1.0 / 21, np.float32) + _rand((P5, 21), 11, 1e-3)).astype(np.float32)
    out = det.forward(_cu(loc5), _cu(conf5), pri5.to(DEV), arm_loc_data=_cu(arm5), scale=sc).cpu().numpy()
    ref = orc.detect(loc5, conf5, pri5.numpy(), arm5, sc)
    assert np.array_equal(out[..., 0], ref[..., 0])
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("cfg_name,dense", [("704", 5), ("1216", 2)])
def test_detect_beyond_16384_priors_matches_oracle(cfg_name, dense):
    """multi_eval.py:21-24 runs a 320-net at 704 pixels (P = 30855) and a 512-net at 1216 (P = 92055, where the
    score keys no longer fit LDS and live in global memory).  `dense` classes have EVERY prior as a candidate,
    one class is a single huge score tie (ascending prior index decides), the rest are sparse."""
    from tdrn_amd.data import multi_cfg, multi_cfg_512
    cfg = multi_cfg[cfg_name] if cfg_name in multi_cfg else multi_cfg_512[cfg_name]
    pri = PriorBox(cfg).forward()
    P = pri.shape[0]
    assert P == {"704": 30855, "1216": 92055}[cfg_name]
    loc, arm, conf = synth.synth_detect_inputs(1, P, 21, 9.0, seed=6)
    conf = conf.copy()
    rng = np.random.Generator(np.random.PCG64(12))
    for c in range(1, 1 + dense):
        conf[:, c] = (1.0 / 21 + 1e-3 * rng.standard_normal(P)).astype(np.float32)
    conf[:, 1 + dense] = np.float32(0.04)                       # P-way tie
    sc = [500.0, 375.0, 500.0, 375.0]
    det = Detect(21, 0, 200, 0.01, 0.45)
    out = det.forward(_cu(loc), _cu(conf), pri.to(DEV), arm_loc_data=_cu(arm), scale=sc).cpu().numpy()
    ref = orc.detect(loc, conf, pri.numpy(), arm, sc)
    assert np.array_equal(out[..., 0], ref[..., 0])
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)
    assert (det.last_counts.cpu().numpy()[0, 1:2 + dense] == 200).all()


def test_nms_beyond_16384_boxes_matches_oracle():
    """n > 16384: the sort keys leave LDS (global bitonic network); keep lists stay bit-exact, both rules."""
    for n, spread, seed in ((16385, 400, 3), (40000, 700, 4)):
        dets = _random_dets(n, spread, seed)
        assert len(np.unique(dets[:, 4])) == n
        for strict in (False, True):
            ref = orc.cpu_nms(dets, 0.45, strict_gt=strict)
            got = nms(dets, 0.45, force_cpu=not strict)
            assert got == ref


def test_detect_coco_class_count_matches_oracle():
    """81 classes (COCO, data/config.py COCO_300): the score transpose tile and the per-(image, class) grid at
    a class count other than 21; D8-like regime plus one dense class."""
    B, P, C = 2, 6375, 81
    loc, arm, conf = synth.synth_detect_inputs(B, P, C, 9.5, seed=21)
    conf = conf.copy()
    conf[:P, 7] = (0.02 + 0.5 * np.random.Generator(np.random.PCG64(4)).random(P)).astype(np.float32)   # every prior a candidate
    pri = PriorBox(mb_cfg["VOC_320"]).forward()
    det = Detect(C, 0, 200, 0.01, 0.45)
    out = det.forward(_cu(loc), _cu(conf), pri.to(DEV), arm_loc_data=_cu(arm)).cpu().numpy()
    ref = orc.detect(loc, conf, pri.numpy(), arm, (320,) * 4, num_classes=C)
    assert out.shape == (B, C, 200, 5)
    assert np.array_equal(out[..., 0], ref[..., 0])
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=1e-6)


def test_detect_empty_and_errors():
    P = 6375
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = Detect(21, 0, 200, 0.5, 0.45)
    conf = np.zeros((P, 21), np.float32)
    conf[:, 0] = 1.0
    out = det.forward(torch.zeros(1, P, 4, device=DEV), _cu(conf), pri)
    assert out.shape == (1, 21, 200, 5) and float(out.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        det.forward(torch.zeros(1, P, 4, device=DEV), _cu(conf[:10]), pri)
    lib = _lib.lib()
    assert lib.tdrn_detect(None, None, None, None, None, 1, P, 21, 200, 0.01, 0.45, None, None, None, 0, None) == -1


# ---------------------------------------------------------------------------------------------
# preprocess and the TRN key-frame driver (rank 2)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,size,rgb", [((375, 500), 320, True), ((480, 640), 320, False), ((300, 300), 512, True),
                                            ((1080, 1920), 320, True), ((320, 320), 320, True)])
def test_preprocess_matches_oracle_bit_exact(shape, size, rgb):
    from tdrn_amd.data import base_transform, BaseTransform, MEANS
    rng = np.random.Generator(np.random.PCG64(shape[0] + size))
    frames = rng.integers(0, 256, (2,) + shape + (3,), dtype=np.uint8)
    ref = orc.base_transform_u8(frames, size, MEANS, rgb)
    got = base_transform(torch.from_numpy(frames).to(DEV), size, MEANS, rgb).cpu().numpy()
    assert got.shape == (2, 3, size, size)
    assert np.array_equal(got, ref)
    one, _, _ = BaseTransform(size, MEANS, rgb)(torch.from_numpy(frames[0]).to(DEV))
    assert np.array_equal(one.cpu().numpy()[0], ref[0])
    if shape == (320, 320):        # identity resize: exactly pixel - mean
        exp = frames.astype(np.float32) - np.asarray(MEANS, np.float32)
        assert np.array_equal(got, exp[..., ::-1].transpose(0, 3, 1, 2) if rgb else exp.transpose(0, 3, 1, 2))


# ---------------------------------------------------------------------------------------------
# DetectOTA (second half): fixture = the reference class run over a synthetic video
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tub", [3, 0])
def test_detect_ota_matches_reference_sequence(golden_dir, tub):
    """layers/functions/detection_ota.py:37-179 over 14 frames: identities are created, carried across frames, expire
    ten frames after their object vanished (class 7), new objects continue the class's counter (class 4), sibling
    anchors become tubelets of their own (class 12).  Scores / boxes / identities per slot equal the reference's."""
    from tdrn_amd.layers import DetectOTA
    g = np.load(os.path.join(golden_dir, "detect_ota.npz"))
    ref = g["tub%d" % tub]
    pri = PriorBox(mb_cfg["VOC_320"]).forward().to(DEV)
    det = DetectOTA(21, 0, 200, 0.01, 0.45, tub=tub, tub_thresh=1.0, tub_generate_score=0.1)
    frames = synth.synth_ota_sequence(int(g["frames"]))
    for t, (loc, conf, arm, feat) in enumerate(frames):
        out = det.forward(_cu(loc), _cu(conf), pri, feature=_cu(feat), arm_loc_data=_cu(arm)).cpu().numpy()
        assert out.shape == ref[t].shape == (1, 21, 200, 6 if tub else 5)
        assert np.array_equal(out[..., 0] > 0, ref[t][..., 0] > 0), "frame %d: slot occupancy" % t
        np.testing.assert_allclose(out[..., :5], ref[t][..., :5], rtol=3e-6, atol=1e-6, err_msg="frame %d" % t)
        if tub:
            assert np.array_equal(out[..., 5], ref[t][..., 5]), "frame %d: identities" % t
    if tub:
        assert [len(d) for d in det.tubelets if len(d)] == [9, 14, 4, 13, 10]
        det.init_tubelets()
        assert not any(det.tubelets)
    with pytest.raises(ValueError):
        DetectOTA(21, 0, 200, 0.01, 0.0)


def test_roi_resample_and_ota_similarity_match_torch():
    """DetectOTA's association arithmetic as the library's own kernels (tdrn_roi_resample, tdrn_ota_similarity) against the torch
    ops the reference uses for it (layers/functions/detection_ota.py:86-99: F.upsample(bilinear, align_corners=True) of the box's
    cell range; box_utils.IoU and cos_similarity, layers/box_utils.py:295-367), computed here with plain PyTorch on the CPU."""
    import torch.nn.functional as F
    lib = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(5))
    Cf, Hf, Wf, S = 24, 20, 20, 7
    feat = torch.from_numpy(rng.standard_normal((1, Cf, Hf, Wf)).astype(np.float32))
    cells = np.array([[0, 0, 20, 20], [3, 4, 4, 5], [2, 7, 9, 8], [5, 1, 6, 19], [10, 10, 17, 13], [19, 19, 20, 20], [0, 12, 13, 20]], np.int32)
    n = cells.shape[0]
    out = torch.empty((n, Cf * S * S), dtype=torch.float32, device=DEV)
    fd, cd = feat[0].contiguous().to(DEV), torch.from_numpy(cells).to(DEV)
    _lib.check(lib.tdrn_roi_resample(_lib.ptr(fd), Cf, Hf, Wf, _lib.ptr(cd), n, S, _lib.ptr(out), _lib.current_stream(DEV)), "roi")
    want = torch.cat([F.interpolate(feat[:, :, y0:y1, x0:x1], (S, S), mode="bilinear", align_corners=True).reshape(1, -1)
                      for x0, y0, x1, y1 in cells.tolist()], 0)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    # similarity: 7 detections against 5 tubelets of 1..4 stored rows
    Fdim = Cf * S * S
    xy = rng.uniform(0, 0.6, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + rng.uniform(0.1, 0.4, (n, 2)).astype(np.float32)], 1)
    lens = [1, 4, 2, 3, 1]
    tubes = []
    for L in lens:
        t = rng.standard_normal((L, 5 + Fdim)).astype(np.float32)
        txy = rng.uniform(0, 0.6, (L, 2)).astype(np.float32)
        t[:, 1:3] = txy
        t[:, 3:5] = txy + rng.uniform(0.1, 0.4, (L, 2)).astype(np.float32)
        tubes.append(torch.from_numpy(t))
    tubes[2][:, 5:] = want[3:4] * 0.5 + 0.01 * tubes[2][:, 5:]                 # (one tubelet that resembles detection 3)
    rows = torch.cat(tubes, 0).to(DEV)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(DEV)
    best = torch.empty(n, dtype=torch.float32, device=DEV)
    arg = torch.empty(n, dtype=torch.int32, device=DEV)
    bd = torch.from_numpy(boxes).to(DEV)
    _lib.check(lib.tdrn_ota_similarity(_lib.ptr(bd), _lib.ptr(out), n, Fdim, _lib.ptr(rows), _lib.ptr(off), len(lens), _lib.ptr(best),
                                       _lib.ptr(arg), _lib.current_stream(DEV)), "sim")
    b = torch.from_numpy(boxes)
    heads = torch.stack([t[0, :5] for t in tubes], 0)
    x1 = torch.maximum(b[:, None, 0], heads[None, :, 1]); y1 = torch.maximum(b[:, None, 1], heads[None, :, 2])
    x2 = torch.minimum(b[:, None, 2], heads[None, :, 3]); y2 = torch.minimum(b[:, None, 3], heads[None, :, 4])
    inter = (x2 - x1).clamp(min=0.0) * (y2 - y1).clamp(min=0.0)
    iou = inter / ((((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[:, None] - inter) + ((heads[:, 3] - heads[:, 1]) * (heads[:, 4] - heads[:, 2]))[None, :])
    cos = torch.stack([((want @ t[:, 5:].t()) / (want.norm(2, dim=1)[:, None] * t[:, 5:].norm(2, dim=1)[None, :])).mean(dim=1) for t in tubes], 1)
    sim = torch.exp(iou) * cos
    wb, wa = sim.max(dim=1)
    np.testing.assert_allclose(best.cpu().numpy(), wb.numpy(), rtol=2e-5, atol=2e-6)
    assert np.array_equal(arg.cpu().numpy(), wa.numpy().astype(np.int32))
    assert int(arg[3]) == 2


def test_nms_topk_is_box_utils_nms():
    """tdrn_nms_topk against a numpy restatement of layers/box_utils.py:229-293 written in the test (no "+1", top_k
    prefilter, IoU <= overlap survives, fp32), incl. the candidate threshold and a degenerate (zero-area) pair."""
    rng = np.random.Generator(np.random.PCG64(31))
    n = 900
    xy = rng.uniform(0, 0.8, (n, 2)).astype(np.float32)
    wh = rng.uniform(0.02, 0.3, (n, 2)).astype(np.float32)
    sc = (rng.permutation(n).astype(np.float32) / np.float32(n)).astype(np.float32)
    dets = np.concatenate([xy, xy + wh, sc[:, None]], 1).astype(np.float32)
    dets[5, :4] = dets[6, :4] = np.float32(0.5)                      # two identical zero-area boxes: IoU = 0/0 = NaN -> suppressed
    dets[5, 4], dets[6, 4] = np.float32(0.9905), np.float32(0.9795)      # (tie-free: equal scores have no defined order)

    def ref_nms(d, overlap, min_score, top_k):
        cand = np.nonzero(d[:, 4] > min_score)[0]
        order = cand[np.argsort(d[cand, 4], kind="stable")][-top_k:] if top_k else cand[np.argsort(d[cand, 4], kind="stable")]
        area = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
        keep = []
        idx = order
        while idx.size:
            i = idx[-1]
            keep.append(int(i))
            idx = idx[:-1]
            if not idx.size:
                break
            w = np.maximum(np.minimum(d[idx, 2], d[i, 2]) - np.maximum(d[idx, 0], d[i, 0]), np.float32(0))
            h = np.maximum(np.minimum(d[idx, 3], d[i, 3]) - np.maximum(d[idx, 1], d[i, 1]), np.float32(0))
            inter = w * h
            with np.errstate(invalid="ignore", divide="ignore"):
                iou = inter / ((area[idx] - inter) + area[i])
            idx = idx[iou <= np.float32(overlap)]
        return keep
    lib = _lib.lib()
    d = _cu(dets)
    keep = torch.empty(n, dtype=torch.int32, device=DEV)
    num = torch.zeros(1, dtype=torch.int32, device=DEV)
    nb = lib.tdrn_nms_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    for overlap, min_score, top_k in ((0.45, 0.01, 200), (0.3, 0.5, 200), (0.6, -1.0, 0), (0.45, 2.0, 200)):
        _lib.check(lib.tdrn_nms_topk(_lib.ptr(d), n, overlap, min_score, top_k, _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), nb, None))
        torch.cuda.synchronize()
        got = keep[: int(num.item())].cpu().tolist()
        assert got == ref_nms(dets, overlap, min_score, top_k), (overlap, min_score, top_k)


def test_nms_topk_classes_equals_per_class_calls():
    """tdrn_nms_topk_classes (all classes of a frame in one launch: DetectOTA's class loop) keeps, class by class, exactly the
    indices tdrn_nms_topk keeps on (boxes, scores[:, c]) -- incl. classes without a candidate and the top_k prefilter."""
    import ctypes as C
    from tdrn_amd import _lib
    lib = _lib.lib()
    rng = np.random.Generator(np.random.PCG64(5))
    n, ncls, top_k = 3000, 7, 50
    xy = rng.uniform(0, 0.8, (n, 2)).astype(np.float32)
    wh = rng.uniform(0.02, 0.3, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = rng.uniform(0, 1, (n, ncls)).astype(np.float32)
    scores[:, 3] = 0.0                                    # a class with no candidate above min_score
    scores[:, 5] *= 0.02                                  # ... and one with a handful
    b, s = _cu(boxes), _cu(scores)
    keep = torch.full((ncls, n), -7, dtype=torch.int32, device=DEV)
    num = torch.full((ncls,), -7, dtype=torch.int32, device=DEV)
    ws = torch.empty(lib.tdrn_nms_topk_classes_workspace_bytes(n, ncls), dtype=torch.uint8, device=DEV)
    _lib.check(lib.tdrn_nms_topk_classes(_lib.ptr(b), _lib.ptr(s), n, ncls, 1, 0.45, 0.01, top_k, _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws),
                                         ws.numel(), _lib.current_stream(DEV)))
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    assert num[0] == -7 and (keep[0] == -7).all()         # rows below first_class are not touched
    ws1 = torch.empty(lib.tdrn_nms_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    for c in range(1, ncls):
        dets = _cu(np.concatenate([boxes, scores[:, c:c + 1]], 1))
        k1 = torch.empty(n, dtype=torch.int32, device=DEV)
        n1 = torch.zeros(1, dtype=torch.int32, device=DEV)
        _lib.check(lib.tdrn_nms_topk(_lib.ptr(dets), n, 0.45, 0.01, top_k, _lib.ptr(k1), _lib.ptr(n1), _lib.ptr(ws1), ws1.numel(),
                                     _lib.current_stream(DEV)))
        cnt = int(n1.item())
        assert num[c] == cnt, c
        assert np.array_equal(keep[c, :cnt], k1.cpu().numpy()[:cnt]), c
    assert num[3] == 0 and 0 < num[5] < num[1]
