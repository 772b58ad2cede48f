"""Condense scripts/profile_config.sh's raw rocprofv3 output into the small files committed under profiles/<dir>:

    python scripts/summarize_config_profile.py gpurun_out/r04_cfg4 profiles/r04_cfg4

  bench.json, bench_traced_graph.json, per_op.txt   as produced
  kernel_stats_graph_replay.csv                      rocprofv3 --stats of the hipGraph-replay process, as is
  traffic_by_kernel.csv                              per kernel name: dispatches and average duration in the PMC runs, HBM-side bytes
                                                     per dispatch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the gfx950 FETCH_SIZE
                                                     correction of MI355X_MICROARCH.md), and the sum over ONE step
  summary.txt                                        the step's totals: kernel time, HBM-side bytes, achieved GB/s, against the line"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    n = name.replace("void tdrn::", "").replace("tdrn::", "")
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).strip()


def rows(d, kind):
    f = glob.glob(os.path.join(d, "**", "*_%s.csv" % kind), recursive=True)
    return list(csv.DictReader(open(max(f, key=os.path.getmtime)))) if f else []      # (gpurun merges into existing directories: the newest run counts)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for f in ("bench.json", "bench_traced_graph.json", "per_op.txt"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    ks = glob.glob(os.path.join(src, "trace_graph", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(max(ks, key=os.path.getmtime), os.path.join(dst, "kernel_stats_graph_replay.csv"))
    line = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
    steps_in_pmc = 1 + 2 + 5 + 2 + 2      # warm-up 1 + timed 2 x 1 rep + forward-only 1 + 5 + the two profiled passes x 2 (bench.py); counted below instead
    per = collections.OrderedDict()
    for kind, col in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        rs = rows(os.path.join(src, kind), "counter_collection")
        for r in rs:
            if r.get("Counter_Name") != col or "tdrn" not in r["Kernel_Name"]:
                continue
            e = per.setdefault(short(r["Kernel_Name"]), {"n": {}, "v": {}, "dur": []})
            e["n"][col] = e["n"].get(col, 0) + 1
            e["v"][col] = e["v"].get(col, 0.0) + float(r["Counter_Value"])
    for r in rows(os.path.join(src, "pmc_fetch"), "kernel_trace"):
        if "tdrn" in r["Kernel_Name"]:
            per.setdefault(short(r["Kernel_Name"]), {"n": {}, "v": {}, "dur": []})["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # forwards in the PMC process: the first-conv kernel (or the conv it is fused into / the stride-2 first conv) starts each one
    first = [k for k in per if k.startswith("first_conv") or k.endswith("true>")]
    n_fwd = max([per[k]["n"].get("FETCH_SIZE", 0) for k in first] or [1])
    if line["config"]["workload"].startswith("BASELINE config #5"):
        # a step = 1 static + 4 temporal forwards, or (batched mode) 1 static + ONE temporal forward over all frames
        n_fwd = max(1, n_fwd // (2 if line.get("trn_mode") == "batched" else 5))
    out = []
    tot_b = tot_us = 0.0
    for k, e in per.items():
        nf, nw = e["n"].get("FETCH_SIZE", 0), e["n"].get("WRITE_SIZE", 0)
        if not nf or not nw:
            continue
        fetch = e["v"]["FETCH_SIZE"] / nf * 1024.0 * 2.0
        write = e["v"]["WRITE_SIZE"] / nw * 1024.0
        dur = sum(e["dur"]) / max(1, len(e["dur"]))
        per_step = nf / float(n_fwd)
        out.append((k, nf, round(per_step, 2), round(dur, 2), int(fetch), int(write), int(fetch + write), round((fetch + write) * per_step / 1e6, 2),
                    round((fetch + write) / (dur * 1e-6) / 1e9, 1) if dur > 0 else 0.0))
        tot_b += (fetch + write) * per_step
        tot_us += dur * per_step
    with open(os.path.join(dst, "traffic_by_kernel.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches_in_pmc_run", "dispatches_per_step", "avg_us_in_pmc_run", "fetch_bytes_x2", "write_bytes", "hbm_bytes_per_dispatch", "hbm_MB_per_step", "GB_per_s"])
        for r in sorted(out, key=lambda r: -r[7]):
            w.writerow(r)
    r = line["roofline"]
    with open(os.path.join(dst, "summary.txt"), "w") as f:
        f.write("%s\n" % line["config"]["workload"])
        f.write("line: %.1f frames/s, %.3f ms per step (forward only %.3f ms); roofline %s: achieved %.1f %s of %.0f = %.3f\n" % (
            line["value"], line["ms_per_step"], line["forward_only_ms_per_step"], r["bound"], r["achieved"], r["unit"], r["peak"], r["frac"]))
        f.write("PMC passes (eager, forward only, %d steps counted): HBM-side traffic %.1f MB per step (2 x FETCH_SIZE + WRITE_SIZE), kernel time %.1f us per step -> %.0f GB/s while kernels run\n" % (
            n_fwd, tot_b / 1e6, tot_us, tot_b / 1e9 / (tot_us * 1e-6) if tot_us else 0.0))
        if r["bound"] == "hbm":
            f.write("algorithmic bytes per step %.1f MB -> counter traffic / algorithmic = %.2f\n" % (r["algorithmic_bytes_per_step"] / 1e6, tot_b / r["algorithmic_bytes_per_step"]))
    print(open(os.path.join(dst, "summary.txt")).read())


if __name__ == "__main__":
    main()
