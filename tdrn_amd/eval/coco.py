"""COCO detection results (evaluate_coco.py:138-165): one dict per detection, xywh boxes with the
reference's '+1' width/height and its rounding through '{:.1f}' / '{:.2f}' text."""
import json

import numpy as np


def coco_results(detections, sizes, img_ids, label_map, det_list=None):
    """detections: (B, C, top_k, 5) rows [score, x1, y1, x2, y2], normalised boxes; sizes[i] = (w, h);
    label_map[j] = COCO category id of class j.  Appends to and returns det_list."""
    det = detections.detach().cpu().numpy() if hasattr(detections, "detach") else np.asarray(detections)
    if det_list is None:
        det_list = []
    B, C = det.shape[:2]
    for i in range(B):
        w, h = sizes[i]
        for j in range(1, C):
            d = det[i, j]
            if d.sum() == 0:
                continue
            d = d[d[:, 0] > 0.0]
            boxes = d[:, 1:5].astype(np.float32, copy=True)
            boxes[:, 0] *= w
            boxes[:, 2] *= w
            boxes[:, 1] *= h
            boxes[:, 3] *= h
            for b, s in zip(boxes, d[:, 0]):
                det_list.append({"image_id": img_ids[i], "category_id": label_map[j],
                                 "bbox": [float("{:.1f}".format(b[0])), float("{:.1f}".format(b[1])),
                                          float("{:.1f}".format(b[2] - b[0] + 1)), float("{:.1f}".format(b[3] - b[1] + 1))],
                                 "score": float("{:.2f}".format(s))})
    return det_list


def write_coco_results(det_list, res_file):
    with open(res_file, "w") as f:
        json.dump(det_list, f)
