"""Deterministic synthetic weights and frames (no network, no datasets, no checkpoints).

The reference's evaluation needs Google-Drive checkpoints and VOC images (README.md:9,
data/config.py:6-12); neither exists here, so tests, golden fixtures and bench.py use
tensors generated from (name, shape, seed) with numpy's PCG64 -- identical in the build
container and on the GPU box.  Unlike default-init weights, the BatchNorm statistics are
non-trivial, so BN folding is actually exercised, and the He-style scaling keeps
activations O(1) through the 13-conv trunk so that fp32/bf16 parity numbers are meaningful.
"""
import zlib

import numpy as np

MEANS_RGB = (123.0, 117.0, 104.0)  # data/__init__.py:7-12 subtracts (104,117,123) in BGR order


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), int(seed)]))


def synth_tensor(name, shape, seed, keys=()):
    """One state_dict entry.  `keys` = all key names (to recognise BatchNorm affine params)."""
    r = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, np.int64)
    if name.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if name.endswith("running_var"):
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    prefix = name.rsplit(".", 1)[0]
    is_bn = (prefix + ".running_mean") in keys
    if name.startswith("L2Norm"):
        base = {"L2Norm_4_3": 10.0, "L2Norm_5_3": 8.0}.get(prefix, 10.0)
        return (base * r.uniform(0.9, 1.1, shape)).astype(np.float32)
    if is_bn:
        if name.endswith("weight"):
            return r.uniform(0.8, 1.2, shape).astype(np.float32)
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4:
        top = name.split(".")[0]
        if top == "up_layers":            # ConvTranspose2d (Cin, Cout, 2, 2): one tap per output
            fan_in = shape[0]
        else:
            fan_in = shape[1] * shape[2] * shape[3]
        gain = 2.0
        if shape[1] == 3 and top == "backbone":
            gain = 2.0 / 75.0 ** 2         # first conv eats raw (pixel - mean) values, rms ~ 75
        if top in ("arm_loc", "arm_conf", "odm_loc", "odm_conf", "odm_loc_2", "odm_conf_2"):
            gain = 0.5                     # heads: O(1) outputs (logits a few units)
        if top in ("odm_loc", "odm_loc_2"):
            gain = 0.1                     # refined box deltas ~ N(0,1)
        if top in ("offset", "offset2"):
            gain = 1.0                     # offsets ~ O(1) pixel: fractional sampling everywhere
        return (np.sqrt(gain / fan_in) * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 1:                    # conv bias
        return (0.05 * r.standard_normal(shape)).astype(np.float32)
    raise ValueError("unexpected parameter %s %r" % (name, shape))


def synth_state_dict(shapes, seed=0):
    """shapes: ordered {key: shape} (e.g. {k: v.shape for k, v in net.state_dict().items()})."""
    keys = set(shapes)
    return {k: synth_tensor(k, s, seed, keys) for k, s in shapes.items()}


def synth_frames(batch, size, seed=0, rgb=True):
    """`VOC-shaped' frames: U(0,255) - mean, fp32 NCHW (data/voc0712.py:460-469 contract)."""
    r = _rng("frames", seed)
    x = r.uniform(0.0, 255.0, (batch, 3, size, size)).astype(np.float32)
    means = MEANS_RGB if rgb else MEANS_RGB[::-1]
    x -= np.asarray(means, np.float32).reshape(1, 3, 1, 1)
    return x


def synth_detect_inputs(batch, num_priors, num_classes=21, bias=8.0, seed=1):
    """Stand-alone Detect inputs of SURVEY.md 8(d): loc, arm_loc = 0.5 N(0,1); logits = N(0,1)
    with +bias on class 0; conf = softmax(logits).  bias 6/8/9 = the D6/D8/D9 regimes."""
    r = _rng("detect", seed)
    loc = (0.5 * r.standard_normal((batch, num_priors, 4))).astype(np.float32)
    arm = (0.5 * r.standard_normal((batch, num_priors, 4))).astype(np.float32)
    logits = r.standard_normal((batch * num_priors, num_classes)).astype(np.float32)
    logits[:, 0] += np.float32(bias)
    e = np.exp(logits - logits.max(1, keepdims=True))
    conf = (e / e.sum(1, keepdims=True)).astype(np.float32)
    return loc, arm, conf


def synth_ota_sequence(n_frames, num_priors=6375, num_classes=21, feat_shape=(1, 16, 40, 40)):
    """A short synthetic video for DetectOTA (tubelet linking): per frame (loc (1,P,4), conf (P,C), arm_loc (1,P,4),
    feature (1,Cf,Hf,Wf)).  Sparse background candidates (D9 regime) plus ten strong, slowly jittering detections in
    each of five classes; class 7 loses six of them after frame 1, class 4 gains four at frame 5, class 12 gets the
    sibling anchors (other aspect ratio, same cell) of three objects from frame 6 on."""
    loc, arm, conf0 = synth_detect_inputs(1, num_priors, num_classes, 9.0, seed=1)
    r = _rng("ota", 5)
    feat0 = r.standard_normal(feat_shape).astype(np.float32)
    strong = {c: r.choice(num_priors // 3, 10, replace=False) * 3 + 1 for c in (1, 4, 7, 12, 20)}     # anchor 1 (ar 2) of a cell
    late4 = r.choice(num_priors // 3, 4, replace=False) * 3 + 1
    base_score = {c: (0.35 + 0.5 * r.random(len(v))).astype(np.float32) for c, v in strong.items()}
    frames = []
    for t in range(n_frames):
        conf = conf0.copy()
        for c, idx in strong.items():
            keep = slice(None) if (t < 2 or c != 7) else slice(0, 4)
            conf[idx[keep], c] = base_score[c][keep] + np.float32(0.002) * r.standard_normal(len(idx[keep])).astype(np.float32)
        if t >= 5:
            conf[late4, 4] = np.float32(0.6) + np.float32(0.01) * np.arange(4, dtype=np.float32)
        if t >= 6:
            conf[strong[12][:3] + 1, 12] = base_score[12][:3] - np.float32(0.05)                          # anchor 2 (ar 1/2)
        l = loc + np.float32(0.003) * r.standard_normal(loc.shape).astype(np.float32)
        f = feat0 + np.float32(0.02) * r.standard_normal(feat_shape).astype(np.float32)
        frames.append((l.astype(np.float32), conf.astype(np.float32), arm, f.astype(np.float32)))
    return frames
