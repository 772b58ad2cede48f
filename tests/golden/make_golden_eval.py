"""Golden vectors for tdrn_amd.eval from the reference's OWN functions (build container only).

evaluate.py / evaluate_coco.py parse command lines and open datasets at import time, so they cannot be
imported; this script reads their text at run time, pulls the function definitions it needs out of the
syntax tree and executes them in a namespace that supplies the module globals they use (`args`,
`labelmap`, `set_type`).  Nothing of the reference is stored: only the inputs fed to, and the outputs
produced by, those functions go into tests/golden/eval_formats.json.

    python tests/golden/make_golden_eval.py
"""
import ast
import json
import os
import pickle
import sys
import tempfile
import types
import xml.etree.ElementTree as ET

import numpy as np

REF = os.environ.get("TDRN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
LABELS = ("aeroplane", "bicycle", "bird")


def reference_functions(path, names, extra):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    ns = dict(extra)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def scenario(seed=11):
    """12 images, 3 classes: ground truth with some difficult boxes; detections = jittered ground truth,
    duplicates and clutter, as the (C, top_k, 5) rows Detect emits (normalised boxes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    names = ["%06d" % (i + 1) for i in range(12)]
    sizes = [(int(rng.integers(320, 501)), int(rng.integers(240, 376))) for _ in names]
    recs = {}
    for n, (w, h) in zip(names, sizes):
        objs = []
        for _ in range(int(rng.integers(0, 5))):
            x1, y1 = int(rng.integers(0, w - 60)), int(rng.integers(0, h - 60))
            bw, bh = int(rng.integers(30, 200)), int(rng.integers(30, 150))
            objs.append({"name": LABELS[int(rng.integers(0, 3))], "pose": "Unspecified", "truncated": 0,
                         "difficult": int(rng.random() < 0.2),
                         "bbox": [x1, y1, min(x1 + bw, w - 1), min(y1 + bh, h - 1)]})
        recs[n] = objs
    top_k = 8
    det = np.zeros((len(names), len(LABELS) + 1, top_k, 5), np.float32)
    for i, (n, (w, h)) in enumerate(zip(names, sizes)):
        for j, cls in enumerate(LABELS, start=1):
            rows = []
            for o in recs[n]:
                if o["name"] == cls and rng.random() < 0.8:
                    for _ in range(1 + int(rng.random() < 0.3)):           # sometimes a duplicate
                        jit = rng.normal(0, 6, 4)
                        b = np.array(o["bbox"], np.float64) + jit
                        rows.append([rng.uniform(0.3, 0.99), b[0] / w, b[1] / h, b[2] / w, b[3] / h])
            for _ in range(int(rng.integers(0, 3))):                        # clutter
                x1, y1 = rng.uniform(0, 0.6), rng.uniform(0, 0.6)
                rows.append([rng.uniform(0.02, 0.6), x1, y1, x1 + rng.uniform(0.1, 0.4), y1 + rng.uniform(0.1, 0.4)])
            rows = sorted(rows, key=lambda r: -r[0])[:top_k]
            for k, r in enumerate(rows):
                det[i, j, k] = r
    return names, sizes, recs, det


class Dets(np.ndarray):
    """`dets == []` (evaluate.py:178) was False for a non-empty array under the numpy the reference was
    written for; numpy 2 raises on the broadcast.  Same answer, old behaviour."""
    def __eq__(self, other):
        if isinstance(other, list) and not other:
            return False
        return np.ndarray.__eq__(self, other)
    __hash__ = None


def write_xml(path, objs):
    root = ET.Element("annotation")
    for o in objs:
        e = ET.SubElement(root, "object")
        ET.SubElement(e, "name").text = o["name"]
        ET.SubElement(e, "pose").text = o["pose"]
        ET.SubElement(e, "truncated").text = str(o["truncated"])
        ET.SubElement(e, "difficult").text = str(o["difficult"])
        bb = ET.SubElement(e, "bndbox")
        for tag, v in zip(("xmin", "ymin", "xmax", "ymax"), o["bbox"]):
            ET.SubElement(bb, tag).text = str(v + 1)                         # files are 1-based
    ET.ElementTree(root).write(path)


def main():
    import torch
    if not hasattr(np, "bool"):
        np.bool = bool                                                       # the reference predates numpy 1.24
    names, sizes, recs, det = scenario()
    args = types.SimpleNamespace(dataset_name="VOC0712", set_file_name="test")
    ns = reference_functions(os.path.join(REF, "evaluate.py"),
                             {"parse_rec", "get_voc_results_file_template", "write_voc_results_file", "voc_ap", "voc_eval"},
                             dict(np=np, os=os, pickle=pickle, ET=ET, args=args, labelmap=LABELS, set_type="test"))
    out = {"labels": LABELS, "names": names, "sizes": sizes, "recs": recs, "detections": det.tolist()}
    # all_boxes exactly as evaluate.py:467-482 builds it (its loop body, run on torch CPU tensors)
    all_boxes = [[[] for _ in names] for _ in range(len(LABELS) + 1)]
    dt = torch.from_numpy(det.copy())
    for i, (w, h) in enumerate(sizes):
        for j in range(1, dt.size(1)):
            dets = dt[i, j, :]
            if dets.sum() == 0:
                continue
            mask = dets[:, 0].gt(0.).expand(dets.size(-1), dets.size(0)).t()
            dets = torch.masked_select(dets, mask).view(-1, dets.size(-1))
            boxes = dets[:, 1:]
            boxes[:, 0] *= w
            boxes[:, 2] *= w
            boxes[:, 1] *= h
            boxes[:, 3] *= h
            scores = dets[:, 0].cpu().numpy()
            all_boxes[j][i] = np.hstack((boxes.cpu().numpy(), scores[:, np.newaxis])).astype(np.float32, copy=False)
    out["all_boxes"] = [[(a.tolist() if not isinstance(a, list) else []) for a in row] for row in all_boxes]
    with tempfile.TemporaryDirectory() as tmp:
        dataset = types.SimpleNamespace(ids=[("root", n) for n in names])
        ns["write_voc_results_file"]([[a if isinstance(a, list) else a.view(Dets) for a in row] for row in all_boxes], dataset, tmp)
        out["result_files"] = {c: open(ns["get_voc_results_file_template"]("test", c, tmp)).read() for c in LABELS}
        anno = os.path.join(tmp, "anno")
        os.makedirs(anno)
        for n in names:
            write_xml(os.path.join(anno, n + ".xml"), recs[n])
        out["parsed"] = {n: ns["parse_rec"](os.path.join(anno, n + ".xml")) for n in names[:3]}
        setfile = os.path.join(tmp, "test.txt")
        open(setfile, "w").write("\n".join(names) + "\n")
        out["voc_eval"] = {}
        for use07 in (True, False):
            for c in LABELS:
                cache = os.path.join(tmp, "cache%d" % use07)
                rec, prec, ap = ns["voc_eval"](ns["get_voc_results_file_template"]("test", c, tmp), os.path.join(anno, "%s.xml"),
                                               setfile, c, cache, ovthresh=0.5, use_07_metric=use07)
                out["voc_eval"]["%s|%d" % (c, use07)] = {"rec": np.asarray(rec).tolist(), "prec": np.asarray(prec).tolist(), "ap": float(ap)}
        # an empty results file
        empty = os.path.join(tmp, "results", "comp4_det_test_empty.txt")
        open(empty, "w").close()
        recs_e = dict(recs)
        rec, prec, ap = ns["voc_eval"](os.path.join(tmp, "results", "comp4_det_test_{:s}.txt").replace("{:s}", "empty"),
                                       os.path.join(anno, "%s.xml"), setfile, "bird", os.path.join(tmp, "cache1"))
        out["voc_eval_empty"] = [rec, prec, ap]
    # voc_ap on random curves
    rng = np.random.Generator(np.random.PCG64(5))
    out["voc_ap"] = []
    for _ in range(6):
        n = int(rng.integers(1, 40))
        rec = np.sort(rng.random(n)) * rng.uniform(0.3, 1.0)
        prec = rng.random(n)
        out["voc_ap"].append({"rec": rec.tolist(), "prec": prec.tolist(),
                              "ap07": float(ns["voc_ap"](rec, prec, True)), "ap": float(ns["voc_ap"](rec, prec, False))})
    # COCO entries: the reference builds them inline (evaluate_coco.py:150-160); the statement is run as is
    src = open(os.path.join(REF, "evaluate_coco.py")).read()
    tree = ast.parse(src)
    appends = [n for n in ast.walk(tree) if isinstance(n, ast.Expr) and isinstance(n.value, ast.Call)
               and isinstance(n.value.func, ast.Attribute) and n.value.func.attr == "append"
               and isinstance(n.value.func.value, ast.Name) and n.value.func.value.id == "det_list"]
    assert len(appends) == 1
    stmt = compile(ast.Module(body=[appends[0]], type_ignores=[]), "evaluate_coco.py", "exec")
    label_map = {1: 5, 2: 2, 3: 16}
    det_list = []
    for i, (w, h) in enumerate(sizes[:4]):
        for j in range(1, det.shape[1]):
            d = det[i, j]
            if d.sum() == 0:
                continue
            d = d[d[:, 0] > 0]
            boxes_np = d[:, 1:].copy()
            boxes_np[:, 0] *= w; boxes_np[:, 2] *= w; boxes_np[:, 1] *= h; boxes_np[:, 3] *= h
            for b, s in zip(boxes_np, d[:, 0]):
                exec(stmt, dict(det_list=det_list, img_id=100 + i, label_map=label_map, j=j, b=b, s=s))
    out["coco"] = {"label_map": {str(k): v for k, v in label_map.items()}, "entries": det_list}
    # bbox_vote (multi_eval.py:453-494) on clustered boxes: several near-duplicates per object plus singletons
    nv = reference_functions(os.path.join(REF, "multi_eval.py"), {"bbox_vote"}, dict(np=np))
    out["bbox_vote"] = []
    for case in range(5):
        n_obj = int(rng.integers(1, 6))
        rows = []
        for _ in range(n_obj):
            x1, y1 = rng.uniform(0, 300), rng.uniform(0, 200)
            bw, bh = rng.uniform(20, 150), rng.uniform(20, 120)
            for _ in range(int(rng.integers(1, 6))):
                j = rng.normal(0, 4, 4)
                rows.append([x1 + j[0], y1 + j[1], x1 + bw + j[2], y1 + bh + j[3], rng.uniform(0.05, 0.99)])
        det = np.asarray(rows, np.float32)
        if case == 4:
            det = det[:1]
        out["bbox_vote"].append({"det": det.tolist(), "voted": np.asarray(nv["bbox_vote"](det.copy()), np.float64).tolist()})
    json.dump(out, open(os.path.join(HERE, "eval_formats.json"), "w"))
    print("wrote eval_formats.json: %d result lines, %d coco entries" % (sum(len(v.splitlines()) for v in out["result_files"].values()), len(det_list)))


if __name__ == "__main__":
    sys.exit(main())
