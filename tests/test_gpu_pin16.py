"""The 16-bit kernels pinned from OUTSIDE the repo's own drift measurements.

The bf16 / fp16 plans run kernels the fp32 every-stage test never executes (conv3x3_pp, the FUSE instantiation of
conv3x3_patch, ygemm_k256, deform_sample).  Three kinds of checks, all through the C ABI:

  (a) plan equalities: the default plan against the same net with conv3x3_pp off, with its chained split off -- every output
      bit-identical (full DRN net, 320 px at batch 1 and 32, 512 px at batch 3, both 16-bit types);
  (b) exact-input stage checks: for EVERY conv / conv-transpose / depthwise / pool / L2Norm launch of a plan the stage's own
      materialised input is read back (tdrn_net_read_tensor), the stage is recomputed on the CPU in fp64 with the
      16-bit-rounded BN-folded weights (model/networks.py:136-163 arithmetic, folded as net.hip does), and every output
      element must satisfy
            |got - ref| <= ulp16(|ref| + c S) + c S,     S = sum |x| |w| (+ |bias| + |residual|),  c = 2e-6 (bf16) / 4e-5 (fp16)
      i.e. one rounding of the output to the 16-bit type plus fp32 accumulation noise -- a wrong tile, a dropped tap or
      a stale LDS row is O(|ref|), four orders of magnitude above that.  Kernel bugs and accumulated drift are thereby
      separable: drift lives in the INPUT, which the reference convolution shares.  c is per type: the bf16 matrix-core
      path behaves as exact products + fp32 accumulation (every stage lands within HALF an output ulp + 2e-6 S: measured
      worst error / tolerance 0.50); v_mfma_f32_32x32x16_f16 does not -- its 16-term dot products lose low bits of the
      22-bit fp16 products (measured on gfx950, identically in conv_igemm, conv3x3_patch and conv3x3_pp: up to 2e-5 S, visible
      only on outputs that cancel to |y| << S, e.g. one cout of conv2_1 whose outputs hover at 0.003 under S = 29; fp16
      subnormals are NOT flushed: a flushed reference is worse), so c(fp16) = 4e-5;
  (c) the transform-then-sample deformable heads: ygemm_k256's Y against an fp64 GEMM of its device input with the
      rounded per-tap weights (same bound), and deform_sample's output against a blend of THE DEVICE'S OWN Y rows with
      bilinear weights recomputed from the device's own fp32 offsets by the reference's rule
      (utils/deformconv/deform_conv_cuda_kernel.cu:15-51,189-203), fp32 accumulation noise only.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tdrn_amd import _lib
from tdrn_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_num_threads(min(16, torch.get_num_threads()))      # (the fp64 reference convolutions: the GPU hosts' 256 threads oversubscribe)
C_ACC = {"bf16": 2e-6, "fp16": 4e-5}     # accumulation noise relative to S = sum |x||w| (module docstring)
TORCH16 = {"bf16": torch.bfloat16, "fp16": torch.float16}
UNIT = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}     # unit roundoff of the 16-bit types


def _build(modname, args, phase="test", seed=0, flags=0, dtype="bf16"):
    import importlib
    net = importlib.import_module("tdrn_amd.model." + modname).build_net(phase, *args)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    if phase == "train":
        flags |= _lib.PLAN_TS_ONE_RANGE      # (the stage checks read the whole batch's Y back: not in cache-sized ranges of frames)
    if flags:
        net.set_plan_flags(flags)
    net.set_compute_dtype(dtype)
    return net.to(DEV), sd


def _round16(t, dtype):
    return t.float().to(TORCH16[dtype]).double()


def _ulp16(a, dtype):
    """spacing of the 16-bit type at magnitude a (fp64 tensor >= 0)"""
    mant, emin = (7, -126) if dtype == "bf16" else (10, -14)
    e = torch.floor(torch.log2(a.clamp(min=2.0 ** emin))).clamp(min=emin)
    return torch.pow(torch.tensor(2.0, dtype=torch.float64), e - mant)


def _fold(sd, op):
    """BN-folded fp32 weights and bias of a conv op, as net.hip fold()/pack() compute them (double, cast to float)."""
    w = torch.from_numpy(sd[op["w"] + ".weight"]).double()
    cout = w.shape[0]
    scale = torch.ones(cout, dtype=torch.float64)
    shift = torch.zeros(cout, dtype=torch.float64)
    if op["b"]:
        shift = torch.from_numpy(sd[op["b"] + ".bias"]).double().clone()
    if op["bn"]:
        g, be = torch.from_numpy(sd[op["bn"] + ".weight"]).double(), torch.from_numpy(sd[op["bn"] + ".bias"]).double()
        mu, var = torch.from_numpy(sd[op["bn"] + ".running_mean"]).double(), torch.from_numpy(sd[op["bn"] + ".running_var"]).double()
        scale = g / torch.sqrt(var + 1e-5)
        shift = (shift - mu) * scale + be
    wf = (w * scale.view(-1, *([1] * (w.dim() - 1)))).float()
    return wf, shift.float()


def _add(a, b):
    return b if a is None else (a if b is None else a + b)


def _assert_stage(name, got, ref, S, dtype, out16=True, extra=None, report=None):
    got, ref, S = got.double(), ref.double(), S.double()
    tol = C_ACC[dtype] * S
    if extra is not None:
        tol = tol + extra
    if out16:
        tol = tol + _ulp16(ref.abs() + tol, dtype)
    err = (got - ref).abs()
    worst = float((err / tol.clamp(min=1e-30)).max())
    if report is not None:
        report.append((name, worst, float(err.max()), float(ref.abs().max())))
    bad = err > tol
    assert not bool(bad.any()), "%s: %d of %d elements outside one 16-bit rounding + fp32 noise (worst %.1fx, max |err| %.3g at |ref| max %.3g)" % (
        name, int(bad.sum()), bad.numel(), worst, float(err.max()), float(ref.abs().max()))


def _level_offsets(fm):
    offs = [0]
    for f in fm:
        offs.append(offs[-1] + f * f * 3)
    return offs


def _head_map(t, b, level, fm, per):
    """(B,P,per/3) head output -> the (per, H, W) map of one image and pyramid level"""
    offs = _level_offsets(fm)
    f = fm[level]
    return t[b, offs[level]:offs[level + 1]].reshape(f, f, per).permute(2, 0, 1)


def check_stages(net, sd, x, dtype, images, skip_first_input=False, forward=None):
    """(b) + (c) for every launch of net's plan after net(x); `images`: batch rows that are recomputed on the CPU.
    `forward`: how to run the net when net(x) alone does not (the TRN temporal net needs ref_loc)."""
    B = x.shape[0]
    outs = (forward or net)(x)
    torch.cuda.synchronize()
    eng = net._engine
    tinfo = eng.tensor_infos()
    if len(outs) >= 4:
        arm_loc, odm_loc, conf = outs[0], outs[2], outs[3]
    else:                                                    # ssd4scale nets: (loc, conf[, ...]); their heads write arm_loc / conf
        arm_loc, odm_loc, conf = outs[0], outs[0], outs[1]
    fm = eng.fm
    cache = {}

    def tensor(i):
        if i not in cache:
            cache[i] = eng.read_tensor(i, B).cpu()
        return cache[i]

    report, checked = [], {}
    u = UNIT[dtype]
    ops = eng.op_infos()
    for oi, op in enumerate(ops):
        kind = op["kind"]
        name = "%s:%s" % (kind, op["w"] or tinfo[op["in"]][0])
        if kind == "first_conv":
            if oi + 1 < len(ops) and ops[oi + 1]["fused_first"]:
                continue                                       # computed inside the next conv's loader: checked there
            wf, bf = _fold(sd, op)
            w16 = _round16(wf, dtype)
            for b in images:
                xin = _round16(x[b:b + 1].cpu(), dtype)
                y = F.conv2d(xin, w16, bf.double(), stride=op["stride"], padding=1).clamp(min=0)
                S = F.conv2d(xin.abs(), w16.abs(), bf.double().abs(), stride=op["stride"], padding=1)
                _assert_stage(name, tensor(op["out"])[b:b + 1], y, S, dtype, report=report)
        elif kind == "conv":
            if op["w2"]:
                continue                                       # (merged 5x5 + 3x3 heads of refinedet_vgg: not a 16-bit-only kernel)
            wf, bf = _fold(sd, op)
            w16 = _round16(wf, dtype)
            kw = dict(stride=op["stride"], padding=op["pad"], dilation=op["dil"])
            for b in images:
                extra = None
                if op.get("fused_dw"):
                    # computed inside the depthwise op's launch (dwpw.hip): its input is the fp64 restatement of that op on ITS
                    # materialised input, rounded; an intermediate whose rounding falls the other way on the device (a tie
                    # within fp32 noise) differs by one ulp16, bounded by u * S of this stage
                    d = ops[oi - 1]
                    wfd, bfd = _fold(sd, d)
                    xd = tensor(d["in"])[b:b + 1].double()
                    yd = F.conv2d(xd, wfd.double(), bfd.double(), stride=d["stride"], padding=1, groups=xd.shape[1])
                    xin = _round16(yd.clamp(min=0) if d["relu"] else yd, dtype)
                elif op["fused_first"]:
                    # the launch computes conv1_1 itself: its input is the fp64 restatement of that stage, rounded; where the
                    # device's rounding of an intermediate falls the other way (a tie within fp32 noise) the difference is
                    # one ulp16 of that element, bounded by u * S of this stage
                    f0 = ops[oi - 1]
                    wf0, bf0 = _fold(sd, f0)
                    x16 = _round16(x[b:b + 1].cpu(), dtype)
                    xin = _round16(F.conv2d(x16, _round16(wf0, dtype), bf0.double(), stride=f0["stride"], padding=1).clamp(min=0), dtype)
                else:
                    xin = tensor(op["in"])[b:b + 1].double()
                y = F.conv2d(xin, w16, bf.double(), **kw)
                S = F.conv2d(xin.abs(), w16.abs(), bf.double().abs(), **kw)
                if op["fused_first"] or op.get("fused_dw"):
                    extra = _add(extra, 2 * u * S)
                if op["res"] >= 0:
                    r = tensor(op["res"])[b:b + 1].double()
                    y, S = y + r, S + r.abs()
                if op["relu"]:
                    y = y.clamp(min=0)
                if op["out_kind"] != 0:
                    head = {1: arm_loc, 2: odm_loc, 3: conf}[op["out_kind"]].cpu()
                    got = _head_map(head.view(B, eng.num_priors, -1), b, op["level"], fm, 3 * head.shape[-1] if op["out_kind"] == 3 else 12)[None]
                    _assert_stage(name, got, y, S, dtype, out16=False, extra=extra, report=report)
                elif op["pool"] >= 0:
                    # max is 1-Lipschitz in the sup norm: the pooled output is within the window's largest tolerance
                    yp = F.max_pool2d(y, 2, 2)
                    Sp = F.max_pool2d(S, 2, 2)
                    ex = F.max_pool2d(extra, 2, 2) if extra is not None else None
                    _assert_stage(name + "+pool", tensor(op["pool"])[b:b + 1], yp, Sp, dtype, extra=ex, report=report)
                else:
                    _assert_stage(name, tensor(op["out"])[b:b + 1], y, S, dtype, extra=extra, report=report)
        elif kind == "conv_transpose":
            w16 = _round16(torch.from_numpy(sd[op["w"] + ".weight"]), dtype)
            bias = torch.from_numpy(sd[op["b"] + ".bias"]).double() if op["b"] else None
            for b in images:
                xin = tensor(op["in"])[b:b + 1].double()
                y = F.conv_transpose2d(xin, w16, bias, stride=2)
                S = F.conv_transpose2d(xin.abs(), w16.abs(), bias.abs() if bias is not None else None, stride=2)
                if op["res"] >= 0:
                    r = tensor(op["res"])[b:b + 1].double()
                    y, S = y + r, S + r.abs()
                if op["relu"]:
                    y = y.clamp(min=0)
                _assert_stage(name, tensor(op["out"])[b:b + 1], y, S, dtype, report=report)
        elif kind == "depthwise":
            if op.get("fused_dw"):
                continue                                         # its output never leaves LDS: checked through the pointwise conv behind it
            wf, bf = _fold(sd, op)                               # (the depthwise kernel keeps its weights in fp32)
            for b in images:
                xin = tensor(op["in"])[b:b + 1].double()
                C = xin.shape[1]
                y = F.conv2d(xin, wf.double(), bf.double(), stride=op["stride"], padding=1, groups=C)
                S = F.conv2d(xin.abs(), wf.double().abs(), bf.double().abs(), stride=op["stride"], padding=1, groups=C)
                if op["relu"]:
                    y = y.clamp(min=0)
                _assert_stage(name, tensor(op["out"])[b:b + 1], y, S, dtype, report=report)
        elif kind == "maxpool":
            for b in images:
                xin = tensor(op["in"])[b:b + 1]
                want = F.max_pool2d(xin, 2, 2, ceil_mode=bool(op["ceil_mode"]))
                assert torch.equal(tensor(op["out"])[b:b + 1], want), name
            report.append((name, 0.0, 0.0, 0.0))
        elif kind == "l2norm":
            wv = torch.from_numpy(sd[op["w"] + ".weight"]).double().view(1, -1, 1, 1)
            for b in images:
                xin = tensor(op["in"])[b:b + 1].double()
                y = wv * xin / (torch.sqrt((xin * xin).sum(1, keepdim=True)) + 1e-10)
                _assert_stage(name, tensor(op["out"])[b:b + 1], y, y.abs() * 4, dtype, report=report)      # (fp32 norm + divide: a few 1e-7 relative)
        elif kind == "deform_heads" and op["y"] >= 0:
            _check_transform_then_sample(eng, sd, op, tensor, odm_loc.cpu(), conf.cpu(), fm, B, images, dtype, report, name)
        else:
            continue
        checked[kind] = checked.get(kind, 0) + 1
    return report, checked


def _check_transform_then_sample(eng, sd, op, tensor, odm_loc, conf, fm, B, images, dtype, report, name):
    H = W = fm[op["level"]]
    HW, M = H * W, B * H * W
    nc3 = conf.shape[-1] * 3
    ncol = 12 + nc3
    # ---- per-tap weights, taps of the 3x3 branch first, then the 5x5 branch (net.hip pack(): deform_y_col order)
    branches = [(op["w"], op["b"], 3, 1, op["off_c0"][0])]
    if op["n_branches"] == 2:
        branches.append((op["w2"], op["b2"], op["k2"], op["pad2"], op["off_c0"][1]))
    wt = []
    for ln, cn, k, pad, c0 in branches:
        wl = torch.from_numpy(sd[ln + ".weight"]).double()
        wc = torch.from_numpy(sd[cn + ".weight"]).double()
        wcat = torch.cat([wl, wc], 0)                       # (75, 256, k, k)
        wt.append(_round16(wcat.reshape(ncol, wcat.shape[1], k * k).permute(2, 0, 1), dtype))     # (taps, 75, 256)
    wt = torch.cat(wt, 0)
    taps = wt.shape[0]
    # ---- the device's Y, raw
    raw = tensor(op["y"])                                    # (B, ycols, H, W): an NHWC reading of the buffer
    G = max(1, op["y_groups"])                               # column groups of 80 (12 + 3 * classes > 80): one buffer region each
    ycols = raw.shape[1] // G
    flat_all = raw.permute(0, 2, 3, 1).reshape(-1)
    parts = []
    for g in range(G):
        flat = flat_all[g * ycols * M:(g + 1) * ycols * M]
        if op["y_tap_major"]:
            parts.append(flat[:taps * M * 80].reshape(taps, M, 80))
        else:
            rows = flat.reshape(M, ycols)
            parts.append(torch.stack([rows[:, (t // 3) * 256 + (t % 3) * 80:(t // 3) * 256 + (t % 3) * 80 + 80] for t in range(taps)], 0))
    Yd = torch.cat(parts, 2).double()                        # (taps, M, 80 G): column c of group g = output column 80 g + c
    assert G == (ncol + 79) // 80, name
    assert float(Yd[:, :, ncol:].abs().max()) == 0.0, name + ": padding columns of Y are not zero"
    X = tensor(op["in"]).double()                            # (B, 256, H, W)
    off = tensor(op["off"])                                  # (B, Coff, H, W) fp32, exact
    for b in images:
        xb = X[b].reshape(X.shape[1], HW)                    # (256, HW)
        # (c1) ygemm: Y[tap][pixel][c] = sum_k X[pixel][k] W16[c][tap][k]
        yref = torch.einsum("tck,kp->tpc", wt, xb)
        S = torch.einsum("tck,kp->tpc", wt.abs(), xb.abs())
        _assert_stage(name + ":ygemm", Yd[:, b * HW:(b + 1) * HW, :ncol], yref, S, dtype, report=report)
        # (c2) deform_sample: blend of the DEVICE's Y rows; weights by the reference's rule in fp32 from the device's offsets
        ydev = Yd[:, b * HW:(b + 1) * HW, :ncol].numpy()     # (taps, HW, 75)
        out = np.zeros((HW, ncol))
        Sb = np.zeros((HW, ncol))
        near = np.zeros(HW, bool)
        hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        hh, ww = hh.reshape(-1), ww.reshape(-1)
        t0 = 0
        ob = off[b].numpy()
        for ln, cn, k, pad, c0 in branches:
            for t in range(k * k):
                ti, tj = t // k, t % k
                dh = ob[c0 + 2 * t].reshape(-1).astype(np.float32)
                dw = ob[c0 + 2 * t + 1].reshape(-1).astype(np.float32)
                h_in, w_in = hh - pad, ww - pad
                h_im = (h_in + ti).astype(np.float32) + dh
                w_im = (w_in + tj).astype(np.float32) + dw
                ok = (h_im >= 0) & (w_im >= 0) & (h_im < H) & (w_im < W)
                near |= (np.abs(h_im) < 1e-4) | (np.abs(w_im) < 1e-4) | (np.abs(h_im - H) < 1e-4) | (np.abs(w_im - W) < 1e-4)
                h = np.float32(ti) + dh
                w = np.float32(tj) + dw
                height, width = H - h_in, W - w_in
                h_low, w_low = np.floor(h).astype(np.int64), np.floor(w).astype(np.int64)
                ch, cw = h_low >= height - 1, w_low >= width - 1
                h_low = np.where(ch, height - 1, h_low)
                w_low = np.where(cw, width - 1, w_low)
                h_high, w_high = np.where(ch, h_low, h_low + 1), np.where(cw, w_low, w_low + 1)
                h = np.where(ch, h_low.astype(np.float32), h).astype(np.float32)
                w = np.where(cw, w_low.astype(np.float32), w).astype(np.float32)
                lh, lw = h - h_low.astype(np.float32), w - w_low.astype(np.float32)
                eh, ew = np.float32(1) - lh, np.float32(1) - lw
                wgts = [eh * ew, eh * lw, lh * ew, lh * lw]
                r0, r1 = np.clip(h_in + h_low, 0, H - 1), np.clip(h_in + h_high, 0, H - 1)
                q0, q1 = np.clip(w_in + w_low, 0, W - 1), np.clip(w_in + w_high, 0, W - 1)
                for wg, q in zip(wgts, (r0 * W + q0, r0 * W + q1, r1 * W + q0, r1 * W + q1)):
                    wg = np.where(ok, wg, np.float32(0)).astype(np.float64)[:, None]
                    rows = ydev[t0 + t][q]
                    out += wg * rows
                    Sb += wg * np.abs(rows)
            t0 += k * k
        keep = torch.from_numpy(~near)
        got = torch.cat([_head_map(odm_loc, b, op["level"], fm, 12), _head_map(conf, b, op["level"], fm, nc3)], 0).reshape(ncol, HW).t()
        _assert_stage(name + ":sample", got[keep], torch.from_numpy(out)[keep], torch.from_numpy(Sb)[keep] + 1e-30, dtype, out16=False, report=report)
        assert int(near.sum()) < HW // 4


def _print_report(title, report, checked):
    worst = {}
    for n, wv, e, r in report:
        worst[n] = max(worst.get(n, 0.0), wv)
    print("\n%s: stages checked %r; worst error / tolerance per stage:" % (title, checked))
    for n, wv in worst.items():
        print("    %-44s %.3f" % (n, wv))


VGG = ("dualrefinedet_vggbn", (320, 21, 1024, 1, True, True))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("plan", ["default", "no_fuse_first", "igemm_only"])
def test_every_stage_from_its_own_input_batch2(dtype, plan):
    """(b) + (c) on the primary model at batch 2 (flat tiles that span two images; the small-batch kernel choices: 64-cout
    items, conv3x3_patch on the layers conv3x3_pp declines below 192 items), in three plans: the default one (first conv
    fused into conv1_2's loader), the two-launch one (conv1_2 strictly from its materialised input), and the plan without
    the 3x3 direct-conv kernels (every conv on conv_igemm.hip, split-K included).  Raw logits (phase 'train'): the softmax
    would hide the conf head's values."""
    flags = {"default": 0, "no_fuse_first": _lib.PLAN_NO_FUSE_FIRST, "igemm_only": _lib.PLAN_NO_CONV_PATCH}[plan]
    net, sd = _build(VGG[0], VGG[1], phase="train", flags=flags, dtype=dtype)
    x = torch.from_numpy(synth.synth_frames(2, 320, seed=41)).to(DEV)
    report, checked = check_stages(net, sd, x, dtype, images=(0, 1))
    _print_report("%s %s batch 2" % (plan, dtype), report, checked)
    assert checked.get("conv", 0) >= 32 and checked.get("conv_transpose", 0) == 3 and checked.get("deform_heads", 0) == 4
    assert checked.get("l2norm", 0) == 2 and checked.get("maxpool", 0) >= 2
    names = [n for n, _, _, _ in report]
    assert ("first_conv:backbone.0" in names) == (plan != "default")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_every_stage_from_its_own_input_batch32(dtype):
    """The BENCHMARK's launches (batch 32: conv3x3_pp with the chained split on conv3_2 / conv4_x / the 40x40 TCB convs, 128-cout
    items elsewhere, the 512-pixel FUSE items, ygemm_k256 + deform_sample at full size): three of the 32 frames are
    recomputed stage by stage from the device's own stage inputs."""
    net, sd = _build(VGG[0], VGG[1], phase="train", dtype=dtype)
    x = torch.from_numpy(synth.synth_frames(32, 320, seed=43)).to(DEV)
    report, checked = check_stages(net, sd, x, dtype, images=(0, 13, 31))
    _print_report("default %s batch 32" % dtype, report, checked)
    assert checked.get("conv", 0) >= 32 and checked.get("deform_heads", 0) == 4


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_every_stage_512_batch3(dtype):
    """config #3's geometry (64x64 ... 8x8 maps, 2-D tiles everywhere, odd batch)."""
    net, sd = _build(VGG[0], (512,) + VGG[1][1:], phase="train", dtype=dtype)
    x = torch.from_numpy(synth.synth_frames(3, 512, seed=45)).to(DEV)
    report, checked = check_stages(net, sd, x, dtype, images=(2,))
    _print_report("default %s 512 batch 3" % dtype, report, checked)
    assert checked.get("conv", 0) >= 32 and checked.get("deform_heads", 0) == 4


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("plan", ["two_launches", "fused"])
def test_every_stage_mobilenet(dtype, plan):
    """The MobileNet trunk's launches from their own inputs: the default plan (depthwise strips, 1x1 GEMMs on dwpw.hip's
    pw1x1_kernel, stride-2 first conv: every stage strictly) and the opt-in plan TDRN_PLAN_DWPW, where eight conv_dw blocks are ONE
    launch each (the pair is checked from the depthwise op's input; the strict pin of those blocks is the bit-identity of the two
    plans, tests/test_gpu_net.py::test_fused_dwpw_equals_two_launches)."""
    net, sd = _build("dualrefinedet_mobilenet", (320, 21, 1, True), phase="train", dtype=dtype, flags=_lib.PLAN_DWPW if plan == "fused" else 0)
    x = torch.from_numpy(synth.synth_frames(2, 320, seed=47)).to(DEV)
    report, checked = check_stages(net, sd, x, dtype, images=(1,))
    _print_report("mobilenet %s %s" % (plan, dtype), report, checked)
    fused = sum(1 for o in net._engine.op_infos() if o["kind"] == "depthwise" and o["fused_dw"])
    assert fused == (8 if plan == "fused" else 0)
    assert checked.get("depthwise", 0) == 15 - fused and checked.get("conv", 0) >= 30 and checked.get("first_conv", 0) == 1


def _outputs(net, x):
    o = net(x)
    flat = []
    for t in o:
        if torch.is_tensor(t):
            flat.append(t.clone())
        elif t is not None:
            flat.extend(u.clone() for u in t)
    return flat


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_kernel_choice_never_changes_a_bit(dtype):
    """(a) conv3x3_pp == conv3x3_patch and chained split == whole items, on the whole net (what csrc/dev/conv_check.hip shows
    layer by layer in the developer harness): the default plan against TDRN_PLAN_NO_CONV_PP and TDRN_PLAN_NO_PP_SK -- every
    output tensor torch.equal -- at 320 px batch 1 and 32 and at 512 px batch 3.  Not vacuous: the launch lists differ."""
    cases = [(320, 1, 51), (320, 32, 52), (512, 3, 53), (320, 7, 54)]      # (batch 7: ragged unit ranges of conv3x3_ws)
    for size, batch, seed in cases:
        args = (size,) + VGG[1][1:]
        base, _ = _build(VGG[0], args, dtype=dtype)
        x = torch.from_numpy(synth.synth_frames(batch, size, seed=seed)).to(DEV)
        want = _outputs(base, x)
        # (round 5: + the weight-stationary conv3x3_ws.hip on conv1_2 / conv2_1 -- with the first conv fused into its producers, and
        # from a materialised conv1_1 -- against conv3x3_patch.hip's loader / consumer kernel on the same layers)
        for flags in (_lib.PLAN_NO_CONV_PP, _lib.PLAN_NO_PP_SK, _lib.PLAN_NO_CONV_PP | _lib.PLAN_NO_FUSE_FIRST, _lib.PLAN_NO_CONV_WS,
                      _lib.PLAN_NO_CONV_WS | _lib.PLAN_NO_FUSE_FIRST, _lib.PLAN_NO_FUSE_FIRST,
                      _lib.PLAN_NO_YGEMM_V2,         # (+ the transform of the deformable heads on its round-3 schedule)
                      _lib.PLAN_TS_ONE_RANGE,        # (+ the heads over the whole batch at once instead of cache-sized ranges of frames)
                      # (round 6: + conv3x3_patch.hip's tail split -- 64-cout half items in an XCD's last, at most half-filled round --
                      # against whole items: batch 32 cuts conv2_1 / 2_2 / 3_1 / 3_3, batch 7 conv3_x, 512 px batch 3 conv3_x; with
                      # every 3x3 layer on that kernel as well)
                      _lib.PLAN_NO_PATCH_TAIL, _lib.PLAN_NO_PATCH_TAIL | _lib.PLAN_NO_CONV_PP):
            other, _ = _build(VGG[0], args, dtype=dtype, flags=flags)
            got = _outputs(other, x)
            assert len(got) == len(want)
            for u, v in zip(got, want):
                assert torch.equal(u, v), (dtype, size, batch, flags)
            other._engine.check()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_arm_loc_heads_on_head3x3_equal_conv_igemm_up_to_fp32_summation_order(dtype):
    """head3x3.hip (round 5) runs the 12-column fp32 ARM loc heads with the source map staged once per channel chunk instead of once per
    tap; its K order is (chunk, tap, channel) where conv_igemm.hip's is (tap, chunk, channel), so the two agree to fp32 summation
    noise, not bit for bit: the default plan against TDRN_PLAN_NO_HEAD3X3, arm_loc of a VGG and of a MobileNet net, three batches
    (ragged last tiles: 1600 / 400 / 100 / 25 pixels per image).  The exact-input stage checks above bound both against fp64."""
    for model, args in (VGG, ("dualrefinedet_mobilenet", (320, 21, 1, True))):
        for batch, seed in ((1, 61), (5, 62), (32, 63)):
            base, _ = _build(model, args, dtype=dtype)
            other, _ = _build(model, args, dtype=dtype, flags=_lib.PLAN_NO_HEAD3X3)
            x = torch.from_numpy(synth.synth_frames(batch, args[0], seed=seed)).to(DEV)
            a, b = base(x)[0], other(x)[0]                      # arm_loc (B, P, 4)
            assert a.shape == b.shape and torch.isfinite(a).all()
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-6 * max(scale, 1.0) + 1e-6, (model, batch, float((a - b).abs().max()), scale)


def test_lost_handoff_is_an_error_not_a_hang():
    """Fault injection (TDRN_PLAN_FAULT_HANDOFF): the producers of conv3x3_pp's chained split never raise their flag.  The
    consumers' bounded polls run out; the launch ENDS (no hang), the status word is raised, tdrn_net_check reports
    TDRN_E_DEVICE, the next tdrn_net_forward refuses to run -- and the forward after that is clean again."""
    net, _ = _build(VGG[0], VGG[1], dtype="bf16", flags=_lib.PLAN_FAULT_HANDOFF)
    good, _ = _build(VGG[0], VGG[1], dtype="bf16")
    x = torch.from_numpy(synth.synth_frames(32, 320, seed=55)).to(DEV)          # batch 32: the chained split is on
    want = _outputs(good, x)
    good._engine.check()                                                        # (a healthy net reports nothing)
    net(x)
    torch.cuda.synchronize()
    eng = net._engine
    with pytest.raises(_lib.TdrnError) as ei:
        eng.check()
    assert ei.value.code == _lib.E_DEVICE
    eng.check()                                                                 # (the word was cleared by the failing check)
    net(x)
    torch.cuda.synchronize()
    with pytest.raises(_lib.TdrnError) as ei:                                   # the NEXT forward refuses to launch
        net(x)
    assert ei.value.code == _lib.E_DEVICE
    # a batch without the chained split runs clean on the same handle, and equals the healthy net
    x1 = x[:1].contiguous()
    got = _outputs(net, x1)
    torch.cuda.synchronize()
    eng.check()
    for u, v in zip(got, _outputs(good, x1)):
        assert torch.equal(u, v)
    assert len(want) == len(got)


def test_transform_then_sample_batch_ranges_512():
    """the transform-then-sample heads address Y with 32-bit byte offsets; at 512 px a batch of 171
    frames passes 4 GiB at the 64x64 level.  The forward then runs the heads in batch ranges through the same Y buffers:
    it succeeds, and every checked frame equals its single-frame run bit for bit."""
    net, _ = _build(VGG[0], (512,) + VGG[1][1:], dtype="fp16")
    x = torch.from_numpy(synth.synth_frames(4, 512, seed=57)).to(DEV)
    big = x.repeat(44, 1, 1, 1)[:172].contiguous()
    big[171] = big[171].flip(-1)
    arm, offs, odm, conf = net(big)
    torch.cuda.synchronize()
    conf = conf.view(172, -1, 21)
    for b in (0, 170, 171):
        a1, o1, d1, c1 = net(big[b:b + 1])
        assert torch.equal(a1[0], arm[b]) and torch.equal(d1[0], odm[b]) and torch.equal(c1, conf[b]), b
