"""Detections -> result formats and VOC AP: host-side mirrors of
evaluate.py:161-185, 244-426, 467-482 and evaluate_coco.py:138-160; multi-scale / flip testing with box voting
(rank 4, multi_eval.py)."""
from .voc import (collect_all_boxes, get_voc_results_file_template, write_voc_results_file, parse_rec, voc_ap,
                  voc_eval, voc_eval_lines, do_python_eval)
from .coco import coco_results, write_coco_results
from .tta import bbox_vote, scale_filter, merge_detections, MultiScaleTester, MULTI_SCALE

__all__ = ["collect_all_boxes", "get_voc_results_file_template", "write_voc_results_file", "parse_rec", "voc_ap",
           "voc_eval", "voc_eval_lines", "do_python_eval", "coco_results", "write_coco_results", "bbox_vote", "scale_filter", "merge_detections",
           "MultiScaleTester", "MULTI_SCALE"]
