"""Multi-scale + horizontal-flip testing with box voting (multi_eval.py:21-24, 453-494,
513-631).  The frames are resized, mean-subtracted and flipped on the device (`tdrn_preprocess`), both
orientations of a scale run as one batch of 2 through the same weights (one plan per input size,
EngineModule.engine_for), Detect runs on the device; only the per-class merge + vote, which the reference also
does in numpy on a handful of boxes, stays on the host."""
import numpy as np
import torch

from ..data import base_transform

MULTI_SCALE = {'320': [192, 320, 384, 448, 512, 576, 704], '512': [320, 512, 640, 1216]}   # multi_eval.py:21-24

# multi_eval.py:571-624: which boxes of a scale are trusted -- (use max side?, comparison, pixels)
_SIZE_RULES = {
    ('320', 192): ('max', '>', 32), ('512', 320): ('max', '>', 32),
    ('320', 320): ('max', '>', 0), ('512', 512): ('max', '>', 0),
    ('320', 384): ('min', '<', 160), ('512', 640): ('min', '<', 160),
    ('320', 448): ('min', '<', 128),
    ('320', 512): ('min', '<', 96),
    ('320', 576): ('min', '<', 64),
    ('320', 706): ('min', '<', 32), ('512', 1216): ('min', '<', 32),
}


def bbox_vote(det):
    """multi_eval.py:453-494.  det (N, 5) = x1, y1, x2, y2, score.  Repeatedly takes the best remaining box,
    gathers every box with IoU >= 0.45 to it ('+1' convention), and replaces the group by its
    score-weighted mean box carrying the group's best score; singletons pass through."""
    det = np.asarray(det)
    if det.shape[0] <= 1:
        return det
    det = det[det[:, 4].ravel().argsort()[::-1], :]
    out = []
    while det.shape[0] > 0:
        area = (det[:, 2] - det[:, 0] + 1) * (det[:, 3] - det[:, 1] + 1)
        w = np.maximum(0.0, np.minimum(det[0, 2], det[:, 2]) - np.maximum(det[0, 0], det[:, 0]) + 1)
        h = np.maximum(0.0, np.minimum(det[0, 3], det[:, 3]) - np.maximum(det[0, 1], det[:, 1]) + 1)
        inter = w * h
        group = np.where(inter / (area[0] + area[:] - inter) >= 0.45)[0]
        members = det[group, :]
        det = np.delete(det, group, 0)
        if group.shape[0] <= 1:
            out.append(members)
            continue
        weighted = members[:, 0:4] * np.tile(members[:, -1:], (1, 4))
        merged = np.zeros((1, 5))
        merged[:, 0:4] = np.sum(weighted, axis=0) / np.sum(members[:, -1:])
        merged[:, 4] = np.max(members[:, 4])
        out.append(merged)
    # np.row_stack of a float32 first group with float64 merged rows promotes exactly like this
    return np.vstack(out) if out else np.zeros((0, 5))


def scale_filter(ssd_dim, scale, boxes):
    """Row indices of `boxes` (pixels) that multi_eval.py:571-624 keeps for this (net size, test scale)."""
    rule = _SIZE_RULES.get((str(ssd_dim), int(scale)))
    if rule is None:
        return None                                     # the reference has no branch for it (e.g. 320_704)
    side, op, px = rule
    bw, bh = boxes[:, 2] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 1] + 1
    v = np.maximum(bw, bh) if side == 'max' else np.minimum(bw, bh)
    return np.where(v > px if op == '>' else v < px)[0]


def merge_detections(detections_multi, w, h, ssd_dim, num_classes):
    """multi_eval.py:553-631 for one image.  detections_multi: {'<ssd_dim>_<scale>_<flip>': (1, C, top_k, 5)
    array of [score, x1, y1, x2, y2] with normalised boxes}.  Returns {class j: (N, 5) voted [x1,y1,x2,y2,score]}.
    (A scale without a size rule reuses the previous scale's row selection in the reference -- a stale
    `index_temp`; here such a scale is skipped, which is what happens there whenever that selection is empty.)"""
    out = {}
    for j in range(1, num_classes):
        cls_dets = np.zeros((0, 5), np.float32)
        for key, d in detections_multi.items():
            dets = np.asarray(d)[0, j]
            if dets.sum() == 0:
                continue
            dets = dets[dets[:, 0] > 0.0]
            boxes = dets[:, 1:5].astype(np.float32, copy=True)
            if key[-1] == '1':                          # undo the horizontal flip
                x1 = 1 - boxes[:, 0]
                x2 = 1 - boxes[:, 2]
                boxes[:, 0], boxes[:, 2] = x2, x1
            boxes[:, 0] *= w
            boxes[:, 2] *= w
            boxes[:, 1] *= h
            boxes[:, 3] *= h
            keep = scale_filter(ssd_dim, int(key.split('_')[1]), boxes)
            if keep is None or keep.size == 0:
                continue
            part = np.hstack((boxes[keep], dets[keep, 0:1])).astype(np.float32, copy=False)
            cls_dets = part.copy() if cls_dets.size == 0 else np.concatenate((cls_dets, part), axis=0)
        if cls_dets.size != 0:
            voted = bbox_vote(cls_dets)
            if len(voted) != 0:
                out[j] = voted
    return out


class MultiScaleTester(object):
    """net: an EngineModule built for ssd_dim (it plans every other input size on first use); detector: Detect;
    priors: {scale: (P, 4) tensor} from PriorBox(multi_cfg[str(scale)])."""

    def __init__(self, net, detector, priors, ssd_dim=320, mean=(104, 117, 123), scales=None):
        self.net, self.detector, self.priors = net, detector, priors
        self.ssd_dim = int(ssd_dim)
        self.mean = mean
        self.scales = list(scales if scales is not None else MULTI_SCALE[str(ssd_dim)])

    def detect(self, frame_u8):
        """frame_u8: (H, W, 3) uint8 BGR tensor on the GPU.  Returns ({class: voted boxes}, detections_multi)."""
        h, w = int(frame_u8.size(0)), int(frame_u8.size(1))
        pair = torch.stack((frame_u8, torch.flip(frame_u8, dims=[1])))          # cv2.flip(im, 1)
        multi = {}
        for v in self.scales:
            x = base_transform(pair, int(v), self.mean, True)                     # resize, -mean, BGR -> RGB
            r = self.net(x)
            if len(r) == 4:
                arm, _, loc, conf = r
            else:
                (loc, conf), arm = r, None
            det = self.detector.forward(loc, conf, self.priors[int(v)], arm_loc_data=arm).cpu().numpy()
            for flip in (0, 1):
                multi["%d_%d_%d" % (self.ssd_dim, int(v), flip)] = det[flip:flip + 1]
        return merge_detections(multi, w, h, self.ssd_dim, self.detector.num_classes), multi
