"""Seeded fuzz of the Detect / NMS device path against the CPU oracle (checker): ragged prior counts around every
internal boundary (wave, chunk of 1024 candidates, the 16384-prior LDS limit), heavy score ties that straddle chunk
boundaries, every candidate density from none to all, small and large top_k.  Bit-exact slot occupancy and scores."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tdrn_amd.layers import Detect
from tdrn_amd.utils.nms_wrapper import nms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _priors(rng, P):
    c = rng.random((P, 2)).astype(np.float32)
    wh = (0.02 + 0.3 * rng.random((P, 2))).astype(np.float32)
    return np.concatenate([c, wh], 1)


def _case(seed):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    P = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1023, 1024, 1025, 2047, 2049, 3000, 6375, 16383, 16384, 16385, 20001]))
    if seed % 5 == 4:
        P = int(rng.integers(1, 9000))
    C = int(rng.choice([2, 3, 5, 21]))
    B = int(rng.choice([1, 2, 3]))
    top_k = int(rng.choice([1, 5, 200, 400]))
    conf_thresh = float(rng.choice([0.01, 0.05, 0.3]))
    nms_thresh = float(rng.choice([0.3, 0.45, 0.7]))
    levels = int(rng.choice([0, 0, 3, 17, 256]))            # 0 = continuous scores, else that many distinct values (ties)
    density = float(rng.choice([0.0, 0.01, 0.2, 1.0]))       # share of priors above the threshold per class
    loc = (0.5 * rng.standard_normal((B, P, 4))).astype(np.float32)
    arm = (0.5 * rng.standard_normal((B, P, 4))).astype(np.float32) if seed % 3 else None
    conf = (conf_thresh * rng.random((B * P, C))).astype(np.float32) * np.float32(0.999)
    hot = rng.random((B * P, C)) < density
    s = conf_thresh + (1.0 - conf_thresh) * rng.random((B * P, C))
    if levels:
        s = conf_thresh + (1.0 - conf_thresh) * (np.floor(rng.random((B * P, C)) * levels) + 1) / (levels + 1)
    conf = np.where(hot, s.astype(np.float32), conf).astype(np.float32)
    if seed % 7 == 0:
        conf[:, C - 1] = np.float32(conf_thresh)             # exactly AT the threshold: strict '>' keeps none
    return dict(P=P, C=C, B=B, top_k=top_k, conf_thresh=conf_thresh, nms_thresh=nms_thresh, loc=loc, arm=arm, conf=conf,
                pri=_priors(rng, P), scale=[float(rng.choice([1.0, 320.0, 500.0])), 375.0, 500.0, 375.0])


@pytest.mark.parametrize("seed", range(40))
def test_detect_fuzz_matches_oracle(seed):
    k = _case(seed)
    det = Detect(k["C"], 0, k["top_k"], k["conf_thresh"], k["nms_thresh"])
    t = lambda a: None if a is None else torch.from_numpy(a).to(DEV)
    out = det.forward(t(k["loc"]), t(k["conf"]), t(k["pri"]), arm_loc_data=t(k["arm"]), scale=k["scale"]).cpu().numpy()
    ref, counts = orc.detect(k["loc"], k["conf"], k["pri"], k["arm"], k["scale"], num_classes=k["C"], top_k=k["top_k"],
                             conf_thresh=k["conf_thresh"], nms_thresh=k["nms_thresh"], return_counts=True)
    assert out.shape == ref.shape
    assert np.array_equal(out[..., 0], ref[..., 0]), "scores / slot occupancy differ (P=%d C=%d top_k=%d)" % (k["P"], k["C"], k["top_k"])
    np.testing.assert_allclose(out, ref, rtol=3e-6, atol=2e-6 * max(k["scale"]))
    assert np.array_equal(det.last_counts.cpu().numpy(), np.minimum(counts, k["top_k"]))    # oracle: survivors before the top_k cut


@pytest.mark.parametrize("seed", range(12))
def test_nms_fuzz_matches_oracle(seed):
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    n = int(rng.choice([1, 2, 64, 65, 1000, 4095, 4097, 16384, 16385, 17000]))
    spread = float(rng.choice([50.0, 300.0, 2000.0]))
    xy = (spread * rng.random((n, 2))).astype(np.float32)
    wh = (5.0 + 60.0 * rng.random((n, 2))).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / np.float32(n) + np.float32(0.001)      # tie-free
    dets = np.concatenate([xy, xy + wh, scores[:, None]], 1).astype(np.float32)
    thr = float(rng.choice([0.3, 0.5, 0.7]))
    for strict in (False, True):
        assert nms(dets, thr, force_cpu=not strict) == orc.cpu_nms(dets, thr, strict_gt=strict)


@pytest.mark.parametrize("seed", range(24))
def test_deform_conv_fuzz_matches_oracle(seed):
    """Random shape classes of the deformable op (per-axis kernel / stride / padding / dilation, groups, ragged channels,
    offsets large enough to leave the map on every side) against deform_conv_cuda_kernel.cu's restatement, fp32."""
    from tdrn_amd.model.networks import conv_offset2d
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    G = int(rng.choice([1, 1, 2, 4, 8]))
    Cin = G * int(rng.choice([1, 2, 3, 8, 16, 32]))
    Cout = int(rng.choice([1, 4, 12, 63, 75, 96, 130]))
    kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    sh, sw = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    dh, dw = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    ph, pw = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    N = int(rng.integers(1, 4))
    H = int(rng.integers(dh * (kh - 1) + 1, dh * (kh - 1) + 14))
    W = int(rng.integers(dw * (kw - 1) + 1, dw * (kw - 1) + 14))
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, kh, kw)) * (Cin * kh * kw) ** -0.5).astype(np.float32)
    off = (float(rng.choice([0.0, 0.5, 2.0, 6.0])) * rng.standard_normal((N, G * 2 * kh * kw, Ho, Wo))).astype(np.float32)
    ref = orc.deform_conv_forward(x, off, w, (sh, sw), (ph, pw), (dh, dw), G)
    t = lambda a: torch.from_numpy(a).to(DEV)
    got = conv_offset2d(t(x), t(off), t(w), (sh, sw), (ph, pw), (dh, dw), G).cpu().numpy()
    assert got.shape == ref.shape == (N, Cout, Ho, Wo)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=3e-5)
