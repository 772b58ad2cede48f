"""DetectOTA: drop-in for layers/functions/detection_ota.py:7-179 -- Detect with online tubelet association for video
(`tub` > 0 links detections across frames by box IoU x ROI-feature cosine similarity and gives them identities).

Split like the reference's own cost profile: what touches every prior runs on the device through the C ABI -- the
two-stage decode (tdrn_decode / tdrn_center_size) and the NMS of ALL classes in ONE launch with box_utils.nms's rule
(tdrn_nms_topk_classes: normalised boxes, no "+1", top_k best candidates only, IoU <= overlap survives) -- while the tubelet
bookkeeping on the handful of survivors (python dictionaries in the reference too) stays host-driven, with its small
tensors on the GPU; the association ARITHMETIC is the library's too: tdrn_roi_resample (the 7x7 bilinear ROI features of all kept
boxes of a frame in one launch) and tdrn_ota_similarity (exp(IoU) x mean cosine against every tubelet, row maximum and argument:
one launch per class that has tubelets) -- torch is left with moving bytes (cat / stack / slicing).  Host round trips per frame: ONE for the counts, kept indices, scores and boxes of all classes; with
tub > 0 ONE more for the similarity decisions of all classes (the reference syncs per class and per box).
Tie rule: equal scores keep the LOWER prior index first (tdrn_nms_topk); box_utils.nms walks an ascending torch.sort from the
end, whose tie order is unspecified -- the fixture (tests/golden/detect_ota.npz) is tie-free at the top_k cut.

State (per class): tubelets[cl] = {identity: [tube, hold]} with tube = (<= tub, 5 + F) rows [score, box, roi feature],
newest first, and hold = frames the tubelet survives without a match (loss_hold_len = 10); ides[cl] = their keys.
"""
import numpy as np
import torch

from ... import _lib
from ..box_utils import center_size, decode


def _nms_topk(dets, overlap, min_score, top_k, ws_cache):
    """layers/box_utils.py:229-293 on the device: dets (n,5) [x1,y1,x2,y2,score] -> (keep indices tensor, count)."""
    lib = _lib.lib()
    n = dets.size(0)
    dev = dets.device
    nb = lib.tdrn_nms_workspace_bytes(n)
    ws = ws_cache.get("ws")
    if ws is None or ws.numel() < nb or ws.device != dev:
        ws = ws_cache["ws"] = torch.empty(nb, dtype=torch.uint8, device=dev)
        ws_cache["keep"] = torch.empty(n, dtype=torch.int32, device=dev)
        ws_cache["num"] = torch.zeros(1, dtype=torch.int32, device=dev)
    keep, num = ws_cache["keep"], ws_cache["num"]
    if keep.numel() < n:
        keep = ws_cache["keep"] = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.tdrn_nms_topk(_lib.ptr(dets), n, float(overlap), float(min_score), int(top_k), _lib.ptr(keep), _lib.ptr(num),
                                 _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)), "tdrn_nms_topk")
    count = int(num.item())
    return keep[:count].long(), count


class Detect(object):
    def __init__(self, num_classes, bkg_label, top_k, conf_thresh, nms_thresh, tub=0, tub_thresh=1.0,
                 tub_generate_score=0.7, device='cuda'):
        self.num_classes = num_classes
        self.background_label = bkg_label
        self.top_k = top_k
        self.nms_thresh = nms_thresh
        if nms_thresh <= 0:
            raise ValueError('nms_threshold must be non negative.')
        self.conf_thresh = conf_thresh
        self.variance = [0.1, 0.2]
        self.tub = tub
        self.device = device
        self.tub_thresh = tub_thresh
        self.loss_hold_len = 10
        self.tub_generate_score = tub_generate_score
        self.tub_feature_size = 7
        self._ws = {}
        self.init_tubelets()

    # ---- tubelet state (detection_ota.py:157-179) --------------------------------------------------------------
    def init_tubelets(self):
        if self.tub > 0:
            self.tubelets = [dict() for _ in range(self.num_classes)]
            self.ides = [None for _ in self.tubelets]
            self.history_max_ides = [-1 for _ in range(self.num_classes)]

    def delete_tubelets(self, cl):
        """One frame has passed for class cl: every tubelet loses one unit of hold time; the expired ones go."""
        gone = []
        for ide, tubelet in self.tubelets[cl].items():
            tubelet[-1] -= 1
            if not tubelet[-1]:
                gone.append(ide)
        for ide in gone:
            del self.tubelets[cl][ide]
        self.ides[cl] = [float(k) for k in self.tubelets[cl].keys()]

    # ---- association pieces (detection_ota.py:86-99, layers/box_utils.py:295-367) on the device ------------------------
    def _roi_features(self, feature, cells_h):
        """:86-93 for all kept boxes of a frame: cells_h (n,4) host int32 [x0,y0,x1,y1] -> (n, Cf*7*7) fp32 on the device."""
        lib = _lib.lib()
        dev = feature.device
        n = cells_h.shape[0]
        fm = feature.reshape(feature.size(-3), feature.size(-2), feature.size(-1)).contiguous().float()
        s7 = self.tub_feature_size
        out = torch.empty((n, fm.size(0) * s7 * s7), dtype=torch.float32, device=dev)
        cells = torch.from_numpy(np.ascontiguousarray(cells_h, np.int32)).to(dev)
        _lib.check(lib.tdrn_roi_resample(_lib.ptr(fm), fm.size(0), fm.size(1), fm.size(2), _lib.ptr(cells), n, s7, _lib.ptr(out),
                                         _lib.current_stream(dev)), "tdrn_roi_resample")
        return out

    def _similarity(self, nms_box, roi, tubes):
        """:95-99: (max over tubelets of exp(IoU) * mean cosine, its index) per detection, on the device."""
        lib = _lib.lib()
        dev = roi.device
        rows = torch.cat([t[0] for t in tubes.values()], 0).contiguous()             # (R, 5 + F), a tubelet's newest row first
        off = np.zeros(len(tubes) + 1, np.int32)
        off[1:] = np.cumsum([t[0].size(0) for t in tubes.values()])
        row_off = torch.from_numpy(off).to(dev)
        n = roi.size(0)
        best = torch.empty(n, dtype=torch.float32, device=dev)
        arg = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(lib.tdrn_ota_similarity(_lib.ptr(nms_box), _lib.ptr(roi), n, roi.size(1), _lib.ptr(rows), _lib.ptr(row_off), len(tubes),
                                           _lib.ptr(best), _lib.ptr(arg), _lib.current_stream(dev)), "tdrn_ota_similarity")
        return best, arg

    def _nms_all_classes(self, boxes, conf_i):
        """box_utils.nms for every class of one frame: ONE launch, ONE host read.  boxes (P,4), conf_i (P,C) on the device ->
        (counts [C] ints, ids (C, top_k) int64 device tensor of kept prior indices, scores (C, top_k) and boxes (C, top_k, 4) on
        the host as numpy, rows past a class's count unspecified)."""
        lib = _lib.lib()
        dev = boxes.device
        P, C = conf_i.size(0), conf_i.size(1)
        nb = lib.tdrn_nms_topk_classes_workspace_bytes(P, C)
        ws = self._ws.get("ws_all")
        if ws is None or ws.numel() < nb or ws.device != dev or self._ws.get("P") != P:
            ws = self._ws["ws_all"] = torch.empty(nb, dtype=torch.uint8, device=dev)
            self._ws["keep_all"] = torch.zeros((C, P), dtype=torch.int32, device=dev)
            self._ws["num_all"] = torch.zeros(C, dtype=torch.int32, device=dev)
            self._ws["P"] = P
        keep, num = self._ws["keep_all"], self._ws["num_all"]
        if P > 16384:                                      # beyond the one-launch limit (multi-scale frames): class by class
            for cl in range(1, C):
                dets = torch.cat((boxes, conf_i[:, cl:cl + 1]), 1).contiguous()
                kk, cnt = _nms_topk(dets, self.nms_thresh, self.conf_thresh, self.top_k, self._ws)
                num[cl] = cnt
                if cnt:
                    keep[cl, :cnt] = kk.int()
        else:
            _lib.check(lib.tdrn_nms_topk_classes(_lib.ptr(boxes), _lib.ptr(conf_i), P, C, 1, float(self.nms_thresh), float(self.conf_thresh),
                                                 int(self.top_k), _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(),
                                                 _lib.current_stream(dev)), "tdrn_nms_topk_classes")
        k = min(self.top_k, P)
        ids = keep[:, :k].long().clamp_(0, P - 1)                      # (rows past the count hold stale indices: clamped, never used)
        sc = conf_i.t().gather(1, ids)                                  # (C, k)
        bx = boxes[ids.reshape(-1)].reshape(C, k, 4)
        packed = torch.cat((num.float()[:, None], sc, bx.reshape(C, k * 4)), 1).cpu().numpy()      # the frame's one host read
        counts = packed[:, 0].astype(np.int64)
        counts[0] = 0
        return counts, ids, packed[:, 1:1 + k], packed[:, 1 + k:].reshape(C, k, 4)

    def _roi_cells(self, box, Hf, Wf):
        """:86-89 on the host: the box's cell range on the feature map"""
        x0 = int(np.clip(np.floor(box[0] * Wf), 0, Wf)); y0 = int(np.clip(np.floor(box[1] * Hf), 0, Hf))
        x1 = int(np.clip(np.ceil(box[2] * Wf), 0, Wf)); y1 = int(np.clip(np.ceil(box[3] * Hf), 0, Hf))
        return x0, y0, x1, y1

    def forward(self, loc_data, conf_data, prior_data, feature=None, arm_loc_data=None):
        """loc (B,P,4), conf (B*P,C), priors (P,4), feature (1,Cf,Hf,Wf) when tub > 0, arm_loc (B,P,4)|None.
        Returns (B, C, top_k, 5) rows [score, box] -- or (1, C, top_k, 6) rows [score, box, identity] when tub > 0
        (identity -1 = not linked).  With tub > 0 the batch must be one frame (the reference's video loop)."""
        _lib.require_cuda(loc_data, "loc_data")
        dev = self._dev = loc_data.device
        num, P, C = loc_data.size(0), prior_data.size(0), self.num_classes
        if self.tub > 0 and num != 1:
            raise ValueError("tubelet linking works on one frame at a time")
        width = 6 if self.tub > 0 else 5
        out_h = np.zeros((num, C, self.top_k, width), np.float32)       # assembled on the host, uploaded once
        conf = conf_data.contiguous().float().view(num, P, C)
        pri = prior_data.to(dev).contiguous().float()
        s7 = self.tub_feature_size
        for i in range(num):
            anchors = pri
            if arm_loc_data is not None:
                anchors = center_size(decode(arm_loc_data[i], pri, self.variance))
            boxes = decode(loc_data[i], anchors, self.variance).contiguous()    # (P,4) normalised
            counts, ids, sc_h, bx_h = self._nms_all_classes(boxes, conf[i].contiguous())
            if self.tub == 0:
                for cl in range(1, C):
                    n = int(counts[cl])
                    if n:
                        out_h[i, cl, :n, 0] = sc_h[cl, :n]
                        out_h[i, cl, :n, 1:5] = bx_h[cl, :n]
                continue
            # ---- tub > 0: roi features and similarities of every class on the device, decisions read back ONCE ----------
            Hf, Wf = feature.size(-2), feature.size(-1)
            feats, sims = {}, {}
            cells, first = [], {}
            for cl in range(1, C):
                n = int(counts[cl])
                if n:
                    first[cl] = len(cells)
                    cells += [self._roi_cells(b, Hf, Wf) for b in bx_h[cl, :n]]
            if cells:
                cells = np.asarray(cells, np.int32)
                if ((cells[:, 2] <= cells[:, 0]) | (cells[:, 3] <= cells[:, 1])).any():
                    raise RuntimeError("a kept box covers no cell of the feature map (the reference's F.upsample raises here too)")
                roi_all = self._roi_features(feature, cells)                   # one launch for the frame
            for cl in range(1, C):
                n = int(counts[cl])
                if n == 0:
                    continue
                roi = roi_all[first[cl]:first[cl] + n]
                feats[cl] = roi
                tubes = self.tubelets[cl]
                if tubes:
                    nms_box = boxes[ids[cl, :n]].contiguous()
                    sims[cl] = self._similarity(nms_box, roi, tubes)
            if sims:
                order = sorted(sims)
                flat = torch.cat([torch.cat((sims[cl][0], sims[cl][1].float())) for cl in order]).cpu().numpy()   # the second host read
            pos = 0
            for cl in range(1, C):
                n = int(counts[cl])
                if n == 0:                                                    # no score above conf_thresh (:70-73)
                    self.delete_tubelets(cl)
                    continue
                nms_score = sc_h[cl, :n]
                identity = np.full(n, -1.0, np.float32)
                tubes = self.tubelets[cl]
                if cl in sims:
                    sim_max = flat[pos:pos + n].copy(); sim_idx = flat[pos + n:pos + 2 * n].astype(np.int64)
                    pos += 2 * n
                    # detections claiming the same tubelet: only the most similar one keeps its claim (:101-110)
                    for t in set(sim_idx.tolist()):
                        rivals = np.nonzero(sim_idx == t)[0]
                        if len(rivals) > 1:
                            best = rivals[int(np.argmax(sim_max[rivals]))]
                            for k in rivals:
                                if k != best:
                                    sim_max[k] = 0.0
                    matched = sim_max > self.tub_thresh
                    if matched.any():
                        keys = list(tubes.keys())
                        identity[matched] = np.asarray([float(keys[j]) for j in sim_idx[matched]], np.float32)
                generate = (identity == -1) & (nms_score > self.tub_generate_score)
                n_new = int(generate.sum())
                if n_new > 0:                                                 # fresh identities continue the class's counter
                    first = 0 if self.history_max_ides[cl] < 0 else int(self.history_max_ides[cl]) + 1
                    identity[generate] = np.arange(first, first + n_new, dtype=np.float32)
                    self.history_max_ides[cl] = first + n_new - 1
                out_h[i, cl, :n, 0] = nms_score
                out_h[i, cl, :n, 1:5] = bx_h[cl, :n]
                out_h[i, cl, :n, 5] = identity
                rows = torch.from_numpy(out_h[i, cl, :n, :5].copy()).to(dev)     # [score, box] of the survivors (H2D, no sync)
                for r, ide in enumerate(identity.tolist()):
                    if ide < 0:
                        continue
                    info = torch.cat((rows[r:r + 1], feats[cl][r:r + 1]), dim=1)   # [score, box, roi feature]
                    key = int(ide)
                    if key in tubes:
                        info = torch.cat((info, tubes[key][0]), 0)[:self.tub]
                    tubes[key] = [info, self.loss_hold_len + 1]
                self.delete_tubelets(cl)
        return torch.from_numpy(out_h).to(dev)

    __call__ = forward
