"""CPU ORACLE -- fp32 restatement of the reference's model forwards (test infrastructure only).

The reference's dense arithmetic (Conv2d / BatchNorm2d / MaxPool2d / ConvTranspose2d /
Softmax) *is* PyTorch's, so this restatement drives torch's CPU fp32 functional ops from a
plain state_dict in the order the reference's `forward` does; the deformable op (CUDA-only in
the reference) comes from oracle/tdrn_oracle.c.  It is pinned against the reference's own
modules imported on CPU (tests/test_oracle_pin.py, tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512]


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _conv(sd, name, x, stride=1, padding=0, dilation=1, groups=1):
    b = _t(sd, name + ".bias") if (name + ".bias") in sd else None
    return F.conv2d(x, _t(sd, name + ".weight"), b, stride, padding, dilation, groups)


def _bn(sd, name, x):
    return F.batch_norm(x, _t(sd, name + ".running_mean"), _t(sd, name + ".running_var"),
                        _t(sd, name + ".weight"), _t(sd, name + ".bias"), False, 0.0, 1e-5)


def _l2norm(sd, name, x):
    """layers/modules/l2norm.py:17-21."""
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10
    return _t(sd, name + ".weight").view(1, -1, 1, 1) * (x / norm)


def deform(x, offset, weight, padding, groups=1):
    """ConvOffset2d.forward, model/networks.py:730-733 (stride 1, dilation 1)."""
    out = orc.deform_conv_forward(x.numpy(), offset.numpy(), np.asarray(weight), 1, padding, 1,
                                  groups)
    return torch.from_numpy(out)


def vgg_trunk(sd, x, bn=True, taps=None):
    """model/networks.py:136-163 walked as model/dualrefinedet_vggbn.py:130-150 does.
    Returns (conv4_3 relu, conv5_3 relu, fc7 relu)."""
    idx = 0
    feats = []
    n_conv = 0
    for v in VGG_CFG:
        if v in ("M", "C"):
            x = F.max_pool2d(x, 2, 2, ceil_mode=(v == "C"))
            if taps is not None:
                taps["pool:" + last] = x
            idx += 1
        else:
            x = _conv(sd, "backbone.%d" % idx, x, padding=1)
            if bn:
                x = _bn(sd, "backbone.%d" % (idx + 1), x)
            x = F.relu(x)
            last = "backbone.%d" % idx
            if taps is not None:
                taps[last] = x
            idx += 3 if bn else 2
            n_conv += 1
            if n_conv in (10, 13):          # conv4_3, conv5_3
                feats.append(x)
    x = F.max_pool2d(x, 2, 2)               # pool5_ds=True
    idx += 1
    x = _conv(sd, "backbone.%d" % idx, x, padding=6, dilation=6)
    if bn:
        x = _bn(sd, "backbone.%d" % (idx + 1), x)
    x = F.relu(x)
    if taps is not None:
        taps["backbone.%d" % idx] = x
    idx += 3 if bn else 2
    x = _conv(sd, "backbone.%d" % idx, x)
    if bn:
        x = _bn(sd, "backbone.%d" % (idx + 1), x)
    x = F.relu(x)
    if taps is not None:
        taps["backbone.%d" % idx] = x
    feats.append(x)
    return feats


def _drn_head(sd, arm_sources, x_last, num_classes, multihead, phase, def_groups=1, taps=None):
    tp = taps if taps is not None else {}
    """The part shared by dualrefinedet_vggbn.py:154-206 and dualrefinedet_mobilenet.py:151-199."""
    arm_loc_list, off1, off2 = [], [], []
    for s, a in enumerate(arm_sources):
        loc_a = _conv(sd, "arm_loc.%d" % s, a, padding=1)
        arm_loc_list.append(loc_a.permute(0, 2, 3, 1).contiguous())
        off1.append(_conv(sd, "offset.%d" % s, loc_a))
        if multihead:
            off2.append(_conv(sd, "offset2.%d" % s, loc_a))
        tp["offset.%d" % s] = torch.cat([off1[-1]] + ([off2[-1]] if multihead else []), 1)
    arm_loc = torch.cat([o.view(o.size(0), -1) for o in arm_loc_list], 1)
    # last_layer_trans: conv, ReLU, conv, conv (no ReLU after 2nd/3rd)
    x = _conv(sd, "last_layer_trans.0", x_last, padding=1)
    x = F.relu(x)
    tp["last_layer_trans.0"] = x
    x = _conv(sd, "last_layer_trans.2", x, padding=1)
    tp["last_layer_trans.2"] = x
    x = _conv(sd, "last_layer_trans.3", x, padding=1)
    tp["last_layer_trans.3"] = x
    odm_sources = [x]
    trans = []
    for s in range(3):
        t = _conv(sd, "trans_layers.%d.0" % s, arm_sources[s], padding=1)
        t = F.relu(t)
        tp["trans_layers.%d.0" % s] = t
        trans.append(_conv(sd, "trans_layers.%d.2" % s, t, padding=1))
        tp["trans_layers.%d.2" % s] = trans[-1]
    trans.reverse()
    for i, t in enumerate(trans):
        w = _t(sd, "up_layers.%d.weight" % i)
        b = _t(sd, "up_layers.%d.bias" % i) if ("up_layers.%d.bias" % i) in sd else None
        u = F.relu(F.conv_transpose2d(x, w, b, stride=2) + t)
        tp["up_layers.%d" % i] = u
        x = F.relu(_conv(sd, "latent_layers.%d" % i, u, padding=1))
        tp["latent_layers.%d" % i] = x
        odm_sources.append(x)
    odm_sources.reverse()
    loc_list, conf_list = [], []
    for s, ob in enumerate(odm_sources):
        l = deform(ob, off1[s], sd["odm_loc.%d.weight" % s], 1, def_groups)
        c = deform(ob, off1[s], sd["odm_conf.%d.weight" % s], 1, def_groups)
        if multihead:
            l = l + deform(ob, off2[s], sd["odm_loc_2.%d.weight" % s], 2, def_groups)
            c = c + deform(ob, off2[s], sd["odm_conf_2.%d.weight" % s], 2, def_groups)
        loc_list.append(l.permute(0, 2, 3, 1).contiguous())
        conf_list.append(c.permute(0, 2, 3, 1).contiguous())
    odm_loc = torch.cat([o.view(o.size(0), -1) for o in loc_list], 1)
    odm_conf = torch.cat([o.view(o.size(0), -1) for o in conf_list], 1)
    B = arm_loc.size(0)
    if phase == "test":
        conf = F.softmax(odm_conf.view(-1, num_classes), dim=1)
    else:
        conf = odm_conf.view(B, -1, num_classes)
    return arm_loc.view(B, -1, 4), off1, odm_loc.view(B, -1, 4), conf


def drn_vggbn_forward(sd, x, num_classes=21, bn=True, multihead=False, phase="test",
                      def_groups=1, taps=None):
    """model/dualrefinedet_vggbn.py:119-206."""
    x = torch.as_tensor(x)
    with torch.no_grad():
        tp = taps if taps is not None else {}
        c43, c53, fc7 = vgg_trunk(sd, x, bn, taps)
        srcs = [_l2norm(sd, "L2Norm_4_3", c43), _l2norm(sd, "L2Norm_5_3", c53), fc7]
        tp["L2Norm_4_3"], tp["L2Norm_5_3"] = srcs[0], srcs[1]
        e = _conv(sd, "extras.0", fc7)
        if bn:
            e = F.relu(_bn(sd, "extras.1", e))
            tp["extras.0"] = e
            e = _conv(sd, "extras.3", e, stride=2, padding=1)
            e = F.relu(_bn(sd, "extras.4", e))
            tp["extras.3"] = e
        else:
            e = F.relu(e)
            tp["extras.0"] = e
            e = F.relu(_conv(sd, "extras.2", e, stride=2, padding=1))
            tp["extras.2"] = e
        srcs.append(e)
        return _drn_head(sd, srcs, e, num_classes, multihead, phase, def_groups, taps)


def _conv_dw(sd, name, x, stride):
    """model/networks.py:736-745."""
    c = _t(sd, name + ".0.weight").shape[0]
    x = F.relu(_bn(sd, name + ".1", _conv(sd, name + ".0", x, stride, 1, 1, c)))
    return F.relu(_bn(sd, name + ".4", _conv(sd, name + ".3", x)))


MOBILENET_STRIDES = [1, 2, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1]   # backbone.1 .. backbone.13


def mobilenet_trunk(sd, x):
    """dualrefinedet_mobilenet.py:19-48,136-147 / ssd4scale_mobile.py:20-50,102-111.
    Returns [backbone[:12] out, backbone out, extras.0 out, extras.1 out] (pre-L2Norm)."""
    x = F.relu(_bn(sd, "backbone.0.1", _conv(sd, "backbone.0.0", x, 2, 1)))
    outs = []
    for i, s in enumerate(MOBILENET_STRIDES):
        x = _conv_dw(sd, "backbone.%d" % (i + 1), x, s)
        if i + 1 == 11:
            outs.append(x)
    outs.append(x)
    for k in range(2):
        x = F.relu(_bn(sd, "extras.%d.1" % k, _conv(sd, "extras.%d.0" % k, x)))
        x = _conv_dw(sd, "extras.%d.3" % k, x, 2)
        outs.append(x)
    return outs


def drn_mobilenet_forward(sd, x, num_classes=21, multihead=False, phase="test", def_groups=1, taps=None):
    """model/dualrefinedet_mobilenet.py:127-199."""
    x = torch.as_tensor(x)
    with torch.no_grad():
        a, b, c, d = mobilenet_trunk(sd, x)
        srcs = [_l2norm(sd, "L2Norm_4_3", a), _l2norm(sd, "L2Norm_5_3", b), c, d]
        arm_loc, _, odm_loc, conf = _drn_head(sd, srcs, d, num_classes, multihead, phase,
                                              def_groups, taps)
        return arm_loc, None, odm_loc, conf


def border_rows(taps, multihead, eps=1e-4, def_groups=1):
    """Which prior rows of (odm_loc, conf) may legitimately differ between two fp32 implementations.

    The reference's sampling rule is DISCONTINUOUS where a sample coordinate crosses 0 or the map size
    (deform_conv_cuda_kernel.cu:195: h_im < 0 or h_im >= H -> 0; just inside -> the border pixel's full value), so
    an offset that differs in its last bits flips one output pixel by O(1).  From the ORACLE's offsets
    (taps["offset.<s>"] = [offset ; offset2] of _drn_head) this returns a boolean mask over the B*P prior rows
    (scale-major, then pixel, then anchor) marking the pixels for which some tap of some branch samples within
    `eps` of such a discontinuity; every other row must meet the plain tolerance."""
    masks = []
    s = 0
    while ("offset.%d" % s) in taps:
        off = taps["offset.%d" % s].double().numpy()
        B, _, H, W = off.shape
        hh = np.arange(H, dtype=np.float64).reshape(1, H, 1)
        ww = np.arange(W, dtype=np.float64).reshape(1, 1, W)
        near = np.zeros((B, H, W), bool)
        c0 = 0
        for k, pad in ((3, 1), (5, 2)) if multihead else ((3, 1),):
            for g in range(def_groups):
                for i in range(k):
                    for j in range(k):
                        ch = c0 + g * 2 * k * k + 2 * (i * k + j)
                        h_im = hh - pad + i + off[:, ch]
                        w_im = ww - pad + j + off[:, ch + 1]
                        near |= (np.abs(h_im) < eps) | (np.abs(h_im - H) < eps) | (np.abs(w_im) < eps) | (np.abs(w_im - W) < eps)
            c0 += def_groups * 2 * k * k
        masks.append(np.repeat(near.reshape(B, H * W), 3, axis=1))        # 3 anchors per pixel
        s += 1
    return np.concatenate(masks, axis=1).reshape(-1)


def _ssd4scale_heads(sd, srcs, x, num_classes, phase, deform_on, ref_loc, offset_list, ret_loc, ret_off):
    """Shared by model/ssd4scale_mobile.py:86-140 and model/ssd4scale_vgg.py:71-135 (df_group = 8)."""
    offs = None
    if deform_on:
        offs = offset_list or [_conv(sd, "offset.%d" % s, torch.as_tensor(rl)) for s, rl in enumerate(ref_loc)]
    locs, confs, raw = [], [], []
    for s, src in enumerate(srcs):
        if deform_on:
            l = deform(src, offs[s], sd["arm_loc.%d.weight" % s], 1, 8)
            cf = deform(src, offs[s], sd["arm_conf.%d.weight" % s], 1, 8)
        else:
            l = _conv(sd, "arm_loc.%d" % s, src, padding=1)
            cf = _conv(sd, "arm_conf.%d" % s, src, padding=1)
            raw.append(l)
        locs.append(l.permute(0, 2, 3, 1).contiguous())
        confs.append(cf.permute(0, 2, 3, 1).contiguous())
    B = x.size(0)
    loc = torch.cat([o.view(B, -1) for o in locs], 1).view(B, -1, 4)
    conf = torch.cat([o.view(B, -1) for o in confs], 1)
    conf = F.softmax(conf.view(-1, num_classes), dim=1) if phase == "test" else conf.view(B, -1, num_classes)
    out = [loc, conf]
    if ret_loc:
        out.append(raw)
    if ret_off:
        out.append(offs)
    return tuple(out)


def _vgg_sources(sd, x, bn):
    c43, c53, fc7 = vgg_trunk(sd, x, bn)
    e = _conv(sd, "extras.0", fc7)
    if bn:
        e = F.relu(_bn(sd, "extras.1", e))
        e = F.relu(_bn(sd, "extras.4", _conv(sd, "extras.3", e, stride=2, padding=1)))
    else:
        e = F.relu(e)
        e = F.relu(_conv(sd, "extras.2", e, stride=2, padding=1))
    return [_l2norm(sd, "L2Norm_4_3", c43), _l2norm(sd, "L2Norm_5_3", c53), fc7, e]


def ssd4scale_vgg_forward(sd, x, num_classes=21, phase="test", bn=True, deform_on=False, ref_loc=None,
                          offset_list=None, ret_loc=False, ret_off=False):
    """model/ssd4scale_vgg.py:71-135."""
    x = torch.as_tensor(x)
    with torch.no_grad():
        return _ssd4scale_heads(sd, _vgg_sources(sd, x, bn), x, num_classes, phase, deform_on, ref_loc,
                                offset_list, ret_loc, ret_off)


def refinedet_vgg_forward(sd, x, num_classes=21, use_refine=False, bn=False, multihead=False, phase="test"):
    """model/refinedet_vgg.py:112-219 (plain conv ODM heads; multihead = 3x3 + 5x5 summed)."""
    x = torch.as_tensor(x)
    with torch.no_grad():
        srcs = _vgg_sources(sd, x, bn)
        B = x.size(0)
        arm_loc = None
        if use_refine:
            arm_loc = torch.cat([_conv(sd, "arm_loc.%d" % s, a, padding=1).permute(0, 2, 3, 1).contiguous().view(B, -1)
                                 for s, a in enumerate(srcs)], 1).view(B, -1, 4)
        xx = F.relu(_conv(sd, "last_layer_trans.0", srcs[3], padding=1))
        xx = _conv(sd, "last_layer_trans.3", _conv(sd, "last_layer_trans.2", xx, padding=1), padding=1)
        odm = [xx]
        trans = [_conv(sd, "trans_layers.%d.2" % s, F.relu(_conv(sd, "trans_layers.%d.0" % s, srcs[s], padding=1)), padding=1)
                 for s in range(3)]
        trans.reverse()
        for i, t in enumerate(trans):
            u = F.conv_transpose2d(xx, _t(sd, "up_layers.%d.weight" % i), _t(sd, "up_layers.%d.bias" % i), stride=2)
            xx = F.relu(_conv(sd, "latent_layers.%d" % i, F.relu(u + t), padding=1))
            odm.append(xx)
        odm.reverse()
        locs, confs = [], []
        for s, ob in enumerate(odm):
            l = _conv(sd, "odm_loc.%d" % s, ob, padding=1)
            c = _conv(sd, "odm_conf.%d" % s, ob, padding=1)
            if multihead:
                l = l + _conv(sd, "odm_loc_2.%d" % s, ob, padding=2)
                c = c + _conv(sd, "odm_conf_2.%d" % s, ob, padding=2)
            locs.append(l.permute(0, 2, 3, 1).contiguous().view(B, -1))
            confs.append(c.permute(0, 2, 3, 1).contiguous().view(B, -1))
        odm_loc = torch.cat(locs, 1).view(B, -1, 4)
        conf = torch.cat(confs, 1)
        conf = F.softmax(conf.view(-1, num_classes), dim=1) if phase == "test" else conf.view(B, -1, num_classes)
        return (arm_loc, None, odm_loc, conf) if use_refine else (odm_loc, conf)


def ssd4scale_mobile_forward(sd, x, num_classes=21, phase="test", deform_on=False, ref_loc=None,
                             offset_list=None, ret_loc=False, ret_off=False):
    """model/ssd4scale_mobile.py:86-140 (df_group = 8 when deform)."""
    x = torch.as_tensor(x)
    with torch.no_grad():
        a, b, c, d = mobilenet_trunk(sd, x)
        srcs = [_l2norm(sd, "L2Norm_4_3", a), _l2norm(sd, "L2Norm_5_3", b), c, d]
        return _ssd4scale_heads(sd, srcs, x, num_classes, phase, deform_on, ref_loc, offset_list, ret_loc, ret_off)
