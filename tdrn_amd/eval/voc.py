"""PASCAL-VOC result files and AP, mirroring the reference's evaluate.py.

The reference keeps its configuration in module globals (`args`, `labelmap`, `set_type`); here they are
explicit arguments with the same meaning.  Behaviour kept on purpose:
  * boxes are written 1-based ('+ 1' on every coordinate, evaluate.py:181-185), scores with 3 decimals;
  * the overlap in voc_eval has NO '+1' on widths/heights (evaluate.py:385-391, unlike the NMS), and a
    detection must exceed `ovthresh` strictly;
  * `difficult` flags only exist for VOC0712 (evaluate.py:345-348); other datasets treat every box as easy;
  * with an empty detection file voc_eval returns (0., 0., 0.) (evaluate.py:420-424).
"""
import os
import pickle
import xml.etree.ElementTree as ET

import numpy as np


def collect_all_boxes(detections, sizes, all_boxes=None, first_image=0):
    """evaluate.py:467-482.  detections: (B, C, top_k, 5) rows [score, x1, y1, x2, y2] with normalised
    boxes (what Detect returns when `scale` is left to the caller) or already scaled ones (pass sizes of
    (1, 1)); sizes: per-image (w, h).  Returns all_boxes[cls][image] = (N, 5) float32 [x1, y1, x2, y2, score];
    classes whose rows sum to zero keep the reference's empty list."""
    det = detections.detach().cpu().numpy() if hasattr(detections, "detach") else np.asarray(detections)
    B, C = det.shape[:2]
    if all_boxes is None:
        all_boxes = [[[] for _ in range(first_image + B)] for _ in range(C)]
    for i in range(B):
        w, h = sizes[i]
        for j in range(1, C):                       # j = 0 is the background class
            d = det[i, j]
            if d.sum() == 0:
                continue
            d = d[d[:, 0] > 0.0]
            boxes = d[:, 1:5].astype(np.float32, copy=True)
            boxes[:, 0] *= w
            boxes[:, 2] *= w
            boxes[:, 1] *= h
            boxes[:, 3] *= h
            all_boxes[j][first_image + i] = np.hstack((boxes, d[:, 0:1])).astype(np.float32, copy=False)
    return all_boxes


def get_voc_results_file_template(image_set, cls, output_dir):
    """evaluate.py:161-168."""
    filedir = os.path.join(output_dir, "results")
    if not os.path.exists(filedir):
        os.makedirs(filedir)
    return os.path.join(filedir, "comp4_det_" + image_set + "_%s.txt" % (cls))


def write_voc_results_file(all_boxes, ids, labelmap, set_type, output_dir):
    """evaluate.py:171-185.  ids[i] is the dataset's id tuple; ids[i][1] is the image name."""
    for cls_ind, cls in enumerate(labelmap):
        filename = get_voc_results_file_template(set_type, cls, output_dir)
        with open(filename, "wt") as f:
            for im_ind, index in enumerate(ids):
                dets = all_boxes[cls_ind + 1][im_ind]
                if isinstance(dets, list) and dets == []:
                    continue
                for k in range(dets.shape[0]):
                    f.write("{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n".format(
                        index[1], dets[k, -1], dets[k, 0] + 1, dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))


def parse_rec(filename, dataset_name="VOC0712"):
    """evaluate.py:127-145: PASCAL VOC xml -> list of objects (0-based boxes)."""
    tree = ET.parse(filename)
    objects = []
    for obj in tree.findall("object"):
        o = {"name": obj.find("name").text}
        if dataset_name == "VOC0712":
            o["pose"] = obj.find("pose").text
            o["truncated"] = int(obj.find("truncated").text)
            o["difficult"] = int(obj.find("difficult").text)
        bbox = obj.find("bndbox")
        o["bbox"] = [int(bbox.find("xmin").text) - 1, int(bbox.find("ymin").text) - 1,
                     int(bbox.find("xmax").text) - 1, int(bbox.find("ymax").text) - 1]
        objects.append(o)
    return objects


def voc_ap(rec, prec, use_07_metric=True):
    """evaluate.py:244-275.  VOC07: mean over the recall levels 0, 0.1, ..., 1 of the best precision at
    recall >= level (0 when no point reaches it).  Otherwise: area under the monotone precision envelope."""
    rec, prec = np.asarray(rec), np.asarray(prec)
    if use_07_metric:
        ap = 0.0
        for level in np.arange(0.0, 1.1, 0.1):
            reached = rec >= level
            ap = ap + (np.max(prec[reached]) if np.sum(reached) != 0 else 0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]      # precision envelope, right to left
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def _ground_truth(recs, imagenames, classname, use_difficult):
    """Per image: boxes (n, 4) float, difficult flags, 'already matched' flags; and the count of easy boxes."""
    table, npos = {}, 0
    for name in imagenames:
        objs = [o for o in recs[name] if o["name"] == classname]
        boxes = np.array([o["bbox"] for o in objs]).astype(float)
        hard = np.array([bool(o["difficult"]) if use_difficult else False for o in objs], dtype=bool)
        npos += int((~hard).sum())
        table[name] = (boxes, hard, np.zeros(len(objs), dtype=bool))
    return table, npos


def voc_eval_lines(lines, recs, imagenames, classname, ovthresh=0.5, use_07_metric=True, dataset_name="VOC0712"):
    """The scoring part of voc_eval (evaluate.py:338-424) on the detection file's lines and the parsed
    annotations recs[imagename] = [objects].  Detections are walked in descending confidence
    (np.argsort(-confidence), evaluate.py:372); each claims the ground-truth box of its image with the
    largest overlap if that overlap is > ovthresh: a first claim on an easy box is a true positive, a
    repeated claim a false positive, a claim on a `difficult` box is ignored, anything else is a false
    positive.  Returns (rec, prec, ap), or (0., 0., 0.) for an empty file (evaluate.py:420-424)."""
    table, npos = _ground_truth(recs, imagenames, classname, dataset_name == "VOC0712")
    if not any(lines):
        return 0.0, 0.0, 0.0
    fields = [ln.strip().split(" ") for ln in lines]
    confidence = np.array([float(f[1]) for f in fields])
    boxes = np.array([[float(z) for z in f[2:]] for f in fields])
    order = np.argsort(-confidence)
    nd = len(fields)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for rank, src in enumerate(order):
        gt, hard, taken = table[fields[src][0]]
        bb = boxes[src]
        best, jbest = -np.inf, -1
        if gt.size > 0:
            iw = np.maximum(np.minimum(gt[:, 2], bb[2]) - np.maximum(gt[:, 0], bb[0]), 0.0)     # no '+1' here (evaluate.py:385-386)
            ih = np.maximum(np.minimum(gt[:, 3], bb[3]) - np.maximum(gt[:, 1], bb[1]), 0.0)
            inter = iw * ih
            union = (bb[2] - bb[0]) * (bb[3] - bb[1]) + (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1]) - inter
            ov = inter / union
            jbest = int(np.argmax(ov))
            best = ov[jbest]
        if best > ovthresh:
            if hard[jbest]:
                continue                            # neither tp nor fp
            if taken[jbest]:
                fp[rank] = 1.0
            else:
                tp[rank] = 1.0
                taken[jbest] = True
        else:
            fp[rank] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def voc_eval(detpath, annopath, imagesetfile, classname, cachedir, ovthresh=0.5, use_07_metric=True,
             dataset_name="VOC0712", set_file_name="test"):
    """evaluate.py:278-426 with the reference's file protocol: `detpath.format(classname)` is the results
    file, `annopath % imagename` the xml, annotations are cached in cachedir/annots_<set_file_name>.pkl."""
    if not os.path.isdir(cachedir):
        os.mkdir(cachedir)
    cachefile = os.path.join(cachedir, "annots_" + set_file_name + ".pkl")
    with open(imagesetfile, "r") as f:
        lines = f.readlines()
    if dataset_name == "VID2017":
        lines = [ln.split(" ")[0] for ln in lines]
    imagenames = [x.strip() for x in lines]
    if not os.path.isfile(cachefile):
        recs = {name: parse_rec(annopath % (name), dataset_name) for name in imagenames}
        with open(cachefile, "wb") as f:
            pickle.dump(recs, f)
    else:
        with open(cachefile, "rb") as f:
            recs = pickle.load(f)
    with open(detpath.format(classname), "r") as f:
        det_lines = f.readlines()
    return voc_eval_lines(det_lines, recs, imagenames, classname, ovthresh, use_07_metric, dataset_name)


def do_python_eval(output_dir, labelmap, set_type, annopath, imagesetfile, cachedir, use_07=True,
                   dataset_name="VOC0712", set_file_name="test"):
    """evaluate.py:188-217: per-class AP + <cls>_pr.pkl files; returns (aps, recs, precs, mean AP)."""
    aps, recs, precs = [], [], []
    if not os.path.isdir(output_dir):
        os.mkdir(output_dir)
    for cls in labelmap:
        filename = get_voc_results_file_template(set_type, cls, output_dir)
        rec, prec, ap = voc_eval(filename, annopath, imagesetfile, cls, cachedir, ovthresh=0.5,
                                 use_07_metric=use_07, dataset_name=dataset_name, set_file_name=set_file_name)
        aps.append(ap)
        recs.append(rec)
        precs.append(prec)
        with open(os.path.join(output_dir, cls + "_pr.pkl"), "wb") as f:
            pickle.dump({"rec": rec, "prec": prec, "ap": ap}, f)
    return aps, recs, precs, float(np.mean(aps))
