"""detections -> VOC / COCO result formats and VOC AP, against vectors produced by
the reference's own functions (tests/golden/make_golden_eval.py -> eval_formats.json)."""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from tdrn_amd import eval as ev


@pytest.fixture(scope="module")
def g(golden_dir):
    return json.load(open(os.path.join(golden_dir, "eval_formats.json")))


def _all_boxes(g):
    det = np.asarray(g["detections"], np.float32)
    return det, ev.collect_all_boxes(det, [tuple(s) for s in g["sizes"]])


def test_all_boxes_layout_matches_reference(g):
    det, ab = _all_boxes(g)
    assert len(ab) == det.shape[1] and all(len(row) == det.shape[0] for row in ab)
    for j, row in enumerate(g["all_boxes"]):
        for i, ref in enumerate(row):
            got = ab[j][i]
            if not ref:
                assert isinstance(got, list) and got == []
            else:
                assert got.dtype == np.float32 and np.array_equal(got, np.asarray(ref, np.float32))


def test_voc_result_files_are_byte_identical(g, tmp_path):
    _, ab = _all_boxes(g)
    ev.write_voc_results_file(ab, [("root", n) for n in g["names"]], g["labels"], "test", str(tmp_path))
    for cls, text in g["result_files"].items():
        path = ev.get_voc_results_file_template("test", cls, str(tmp_path))
        assert path.endswith(os.path.join("results", "comp4_det_test_%s.txt" % cls))
        assert open(path).read() == text


def _write_annotations(g, root):
    anno = os.path.join(root, "anno")
    os.makedirs(anno)
    for n in g["names"]:
        top = ET.Element("annotation")
        for o in g["recs"][n]:
            e = ET.SubElement(top, "object")
            ET.SubElement(e, "name").text = o["name"]
            ET.SubElement(e, "pose").text = o["pose"]
            ET.SubElement(e, "truncated").text = str(o["truncated"])
            ET.SubElement(e, "difficult").text = str(o["difficult"])
            bb = ET.SubElement(e, "bndbox")
            for tag, v in zip(("xmin", "ymin", "xmax", "ymax"), o["bbox"]):
                ET.SubElement(bb, tag).text = str(v + 1)
        ET.ElementTree(top).write(os.path.join(anno, n + ".xml"))
    setfile = os.path.join(root, "test.txt")
    open(setfile, "w").write("\n".join(g["names"]) + "\n")
    return os.path.join(anno, "%s.xml"), setfile


def test_parse_rec_and_voc_eval_match_reference(g, tmp_path):
    annopath, setfile = _write_annotations(g, str(tmp_path))
    for n, ref in g["parsed"].items():
        assert ev.parse_rec(annopath % n) == ref
    _, ab = _all_boxes(g)
    ev.write_voc_results_file(ab, [("root", n) for n in g["names"]], g["labels"], "test", str(tmp_path))
    for key, ref in g["voc_eval"].items():
        cls, use07 = key.split("|")
        rec, prec, ap = ev.voc_eval(ev.get_voc_results_file_template("test", cls, str(tmp_path)), annopath, setfile, cls,
                                    os.path.join(str(tmp_path), "cache"), ovthresh=0.5, use_07_metric=bool(int(use07)))
        assert np.array_equal(np.asarray(rec), np.asarray(ref["rec"]))
        assert np.array_equal(np.asarray(prec), np.asarray(ref["prec"]))
        assert ap == ref["ap"]
    # the cache file of the reference's protocol exists and is reused
    assert os.path.isfile(os.path.join(str(tmp_path), "cache", "annots_test.pkl"))
    aps, recs, precs, mean_ap = ev.do_python_eval(str(tmp_path), g["labels"], "test", annopath, setfile,
                                                  os.path.join(str(tmp_path), "cache"), use_07=True)
    assert aps == [g["voc_eval"]["%s|1" % c]["ap"] for c in g["labels"]]
    assert mean_ap == float(np.mean(aps)) and os.path.isfile(os.path.join(str(tmp_path), "bird_pr.pkl"))
    # empty detection file: (0., 0., 0.)
    assert list(ev.voc_eval_lines([], g["recs"], g["names"], "bird")) == g["voc_eval_empty"] == [0.0, 0.0, 0.0]


def test_voc_ap_both_metrics(g):
    for case in g["voc_ap"]:
        rec, prec = np.asarray(case["rec"]), np.asarray(case["prec"])
        assert ev.voc_ap(rec, prec, True) == case["ap07"]
        assert ev.voc_ap(rec, prec, False) == pytest.approx(case["ap"], rel=0, abs=1e-15)


def test_difficult_flags_only_for_voc(g):
    """evaluate.py:345-348: other datasets ignore `difficult`, so recall can only drop or stay."""
    lines = g["result_files"]["bird"].splitlines(True)
    r_voc, _, _ = ev.voc_eval_lines(lines, g["recs"], g["names"], "bird", dataset_name="VOC0712")
    r_vid, _, _ = ev.voc_eval_lines(lines, g["recs"], g["names"], "bird", dataset_name="VID2017")
    n_hard = sum(o["difficult"] for n in g["names"] for o in g["recs"][n] if o["name"] == "bird")
    assert n_hard > 0 and len(r_voc) == len(r_vid)


def test_coco_entries_match_reference(g, tmp_path):
    det = np.asarray(g["detections"], np.float32)[:4]
    label_map = {int(k): v for k, v in g["coco"]["label_map"].items()}
    got = ev.coco_results(det, [tuple(s) for s in g["sizes"][:4]], [100 + i for i in range(4)], label_map)
    assert got == g["coco"]["entries"]
    out = os.path.join(str(tmp_path), "res.json")
    ev.write_coco_results(got, out)
    assert json.load(open(out)) == g["coco"]["entries"]


def test_bbox_vote_matches_reference(g):
    for case in g["bbox_vote"]:
        det = np.asarray(case["det"], np.float32)
        got = np.asarray(ev.bbox_vote(det.copy()), np.float64)
        ref = np.asarray(case["voted"], np.float64).reshape(got.shape)
        assert np.array_equal(got, ref)


def test_merge_detections_unflips_scales_and_filters():
    """multi_eval.py:553-631 on a hand-made pair of views: the flipped view's box maps back onto the plain
    one, the 192-pixel scale drops boxes whose longer side is <= 32 px, and the duplicates are voted."""
    C, K = 3, 4
    plain = np.zeros((1, C, K, 5), np.float32)
    flipped = np.zeros((1, C, K, 5), np.float32)
    small = np.zeros((1, C, K, 5), np.float32)
    plain[0, 1, 0] = [0.9, 0.10, 0.20, 0.50, 0.60]
    flipped[0, 1, 0] = [0.7, 0.50, 0.20, 0.90, 0.60]                       # the same box seen mirrored
    small[0, 1, 0] = [0.8, 0.10, 0.10, 0.15, 0.15]                         # 20 x 15 px at 400 x 300: dropped at 192
    small[0, 2, 0] = [0.6, 0.10, 0.10, 0.60, 0.70]
    multi = {"320_320_0": plain, "320_320_1": flipped, "320_192_0": small}
    out = ev.merge_detections(multi, 400, 300, 320, C)
    assert sorted(out) == [1, 2]
    box = out[1]
    assert box.shape == (1, 5) and box[0, 4] == np.float32(0.9)
    np.testing.assert_allclose(box[0, :4], [40.0, 60.0, 200.0, 180.0], atol=1e-3)
    np.testing.assert_allclose(out[2][0], [40.0, 30.0, 240.0, 210.0, 0.6], atol=1e-4)
    assert ev.scale_filter(320, 704, np.zeros((1, 4), np.float32)) is None     # no branch in the reference
    assert list(ev.scale_filter(512, 1216, np.array([[0, 0, 10, 10], [0, 0, 100, 100]], np.float32))) == [0]
