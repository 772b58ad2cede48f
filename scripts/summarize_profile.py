"""Condense the raw rocprofv3 output of scripts/profile_round.sh into the small files committed under profiles/.

    python scripts/summarize_profile.py gpurun_out/r02_final profiles/r02_final

Writes  kernel_stats.csv          (rocprofv3 --stats of the traced bench run, as is)
        per_layer.csv             one row per launch of a forward: hipEvent time alone / in the step (bench.py --per-op),
                                  kernel-trace duration in the timed loop, TFLOP/s and fraction of the dense MFMA peak, matrix-pipe
                                  busy share, HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
                                  MI355X_MICROARCH.md) against the algorithmic bytes, effective clock
        pmc_by_kernel.csv         the same counters averaged per kernel name
        ../pmc_traffic.json       HBM bytes per launch of the dominant kernel family, read by bench.py
Launches are matched across runs by their order inside a forward (the host enqueues the plan in a fixed order)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

PEAK = 2500.0


def trace_rows(d, kind):
    f = glob.glob(os.path.join(d, "**", "*_%s.csv" % kind), recursive=True)
    return list(csv.DictReader(open(max(f, key=os.path.getmtime)))) if f else []      # (gpurun merges into existing directories: the newest run counts)


def short(name):
    n = name.replace("void tdrn::", "").replace("tdrn::", "")
    return re.sub(r"\(.*", "", n)


def forwards(rows):
    """split the dispatches (enqueue order) into forwards: each starts at the first-conv kernel (or the conv it is fused into)"""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], None
    for r in rows:
        n = short(r["Kernel_Name"])
        if n.startswith("first_conv") or (n.startswith("conv3x3_patch_kernel") and n.rstrip().endswith("true>")) or (n.startswith("conv3x3_ws_kernel") and ", true," in n):     # (16-bit plans: the first conv lives in conv1_2's loader / producers)
            cur = []
            out.append(cur)
        if cur is not None and "tdrn" in r["Kernel_Name"]:
            cur.append(r)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for f in ("bench.json", "bench_traced.json", "per_op.txt"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    ks = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(max(ks, key=os.path.getmtime), os.path.join(dst, "kernel_stats.csv"))
    # ---- per-op event timings -------------------------------------------------------------------------------
    ops = []
    for l in open(os.path.join(src, "per_op.txt")):
        f = l.split()
        if len(f) == 6 and ":" in f[0] and re.match(r"^[0-9.]+$", f[1]):
            ops.append(dict(name=f[0], alone_us=float(f[1]), in_step_us=float(f[2]), gflop=float(f[3]), gbyte=float(f[5])))
    # ---- kernel trace of the timed loop: average duration of the k-th main kernel of a forward ----------------
    tr = forwards(trace_rows(os.path.join(src, "trace"), "kernel_trace"))
    tr = [f for f in tr if len(f) == max(len(g) for g in tr)][-4:]           # steps of the timed loop (full length, incl. Detect)

    def per_family(fwds, value):
        acc = collections.defaultdict(list)
        for f in fwds:
            seen = collections.Counter()
            for r in f:
                n = short(r["Kernel_Name"])
                # (one family: the loader/consumer kernel and the all-waves-compute kernel share the 3x3 layers of a plan)
                fam = "patch" if (n.startswith("conv3x3_patch") or n.startswith("conv3x3_pp") or n.startswith("conv3x3_ws")) else ("igemm" if n.startswith("conv_igemm") else n.split("<")[0])
                acc[(fam, seen[fam])].append(value(r))
                seen[fam] += 1
        return {k: sum(v) / len(v) for k, v in acc.items()}
    dur = per_family(tr, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # the same from the trace of the hipGraph REPLAY (what bench.py times): dispatch ids follow the graph's node order, so the
    # launches are grouped per replay by timestamp order instead
    trg_rows = trace_rows(os.path.join(src, "trace_graph"), "kernel_trace")
    graph_line = None
    if trg_rows:
        trg_rows = [r for r in trg_rows if "tdrn" in r["Kernel_Name"]]
        fam_rows = [r for r in trg_rows if short(r["Kernel_Name"]).startswith(("conv3x3_patch", "conv3x3_pp", "conv3x3_ws"))]
        # the last 3/4 of the dispatches: the timed loop (warm-up and capture passes come first)
        fam_rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        fam_rows = fam_rows[len(fam_rows) // 4:]
        us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in fam_rows]
        if us:
            graph_line = (sum(us) / len(us), len(us))
        ksg = glob.glob(os.path.join(src, "trace_graph", "**", "*_kernel_stats.csv"), recursive=True)
        if ksg:
            shutil.copy(max(ksg, key=os.path.getmtime), os.path.join(dst, "kernel_stats_graph_replay.csv"))
        if os.path.exists(os.path.join(src, "bench_traced_graph.json")):
            shutil.copy(os.path.join(src, "bench_traced_graph.json"), os.path.join(dst, "bench_traced_graph.json"))
    # ---- counters --------------------------------------------------------------------------------------------
    def counters(sub):
        rows = trace_rows(os.path.join(src, sub), "counter_collection")
        by = collections.defaultdict(dict)
        for r in rows:
            by[r["Dispatch_Id"]].update({"Kernel_Name": r["Kernel_Name"], "Dispatch_Id": r["Dispatch_Id"],
                                         r["Counter_Name"]: float(r["Counter_Value"]),
                                         "dur": (int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0))) / 1e3})
        f = forwards(list(by.values()))
        return [g for g in f if len(g) == max(len(h) for h in f)][-1:] if f else []
    fetch, write, sq = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq")
    cf = per_family(fetch, lambda r: r.get("FETCH_SIZE", 0.0))
    cw = per_family(write, lambda r: r.get("WRITE_SIZE", 0.0))
    csq = {k: per_family(sq, lambda r, k=k: r.get(k, 0.0)) for k in
           ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")}
    sq_dur = per_family(sq, lambda r: r.get("dur", 0.0))
    # ---- per-layer table ---------------------------------------------------------------------------------------
    fam_of = lambda name: {"conv3x3_patch_mfma": "patch", "conv_igemm_mfma": "igemm", "first_conv": "first_conv_mfma_kernel",
                           "deform_gemm_mfma": "deform_sample_kernel"}.get(name.split(":")[0])
    seen = collections.Counter()
    rows = [["launch", "alone_us", "in_step_us", "trace_us", "gflop", "tflops_alone", "frac_of_2500_alone", "tflops_in_loop", "frac_in_loop",
             "mfma_busy_share", "hbm_MB", "algorithmic_MB", "eff_clock_ghz"]]
    patch_trace_us = patch_gflop = patch_hbm = 0.0
    n_patch = 0
    for o in ops:
        fam = fam_of(o["name"])
        if fam is None:
            continue
        k = (fam, seen[fam])
        # an igemm op with split-K is two or three dispatches (GEMM + reduce): only single-dispatch families get counters
        single = fam in ("patch", "first_conv_mfma_kernel", "deform_sample_kernel")
        seen[fam] += 1
        t = dur.get(k) if single else None
        tf_a = o["gflop"] / o["alone_us"] * 1e3 if o["alone_us"] else 0.0   # GFLOP / us = PFLOP/s
        tf_l = o["gflop"] / t * 1e3 if t else None
        hbm = (2 * cf.get(k, 0.0) + cw.get(k, 0.0)) * 1024 / 1e6 if single and k in cf else None
        busy = clock = None
        if single and k in csq["SQ_BUSY_CYCLES"] and csq["SQ_BUSY_CYCLES"][k]:
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            gui = csq["GRBM_GUI_ACTIVE"][k] / 8.0
            busy = csq["SQ_VALU_MFMA_BUSY_CYCLES"][k] / (gui * 1024.0) if gui else None
            clock = gui / (sq_dur[k] * 1e3) if sq_dur.get(k) else None
        rows.append([o["name"], "%.1f" % o["alone_us"], "%.1f" % o["in_step_us"], "%.1f" % t if t else "", "%.1f" % o["gflop"],
                     "%.1f" % tf_a, "%.3f" % (tf_a / PEAK), "%.1f" % tf_l if tf_l else "", "%.3f" % (tf_l / PEAK) if tf_l else "",
                     "%.3f" % busy if busy else "", "%.1f" % hbm if hbm is not None else "", "%.1f" % (o["gbyte"] * 1e3),
                     "%.2f" % clock if clock else ""])
        if fam == "patch" and t:
            patch_trace_us += t
            patch_gflop += o["gflop"]
            n_patch += 1
            patch_hbm += hbm or 0.0
    with open(os.path.join(dst, "per_layer.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    # ---- per-kernel counter averages ---------------------------------------------------------------------------
    byk = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for r in trace_rows(os.path.join(src, sub), "counter_collection"):
            if r["Counter_Name"] == key and "tdrn" in r["Kernel_Name"]:
                byk[short(r["Kernel_Name"])][key].append(float(r["Counter_Value"]))
    for r in trace_rows(os.path.join(src, "pmc_sq"), "counter_collection"):
        if "tdrn" in r["Kernel_Name"]:
            byk[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    hdr = ["kernel", "dispatches", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_MB(2*fetch+write)", "wait_any_frac", "wait_inst_any_frac",
           "active_inst_any_frac", "mfma_busy_share_of_1024_simds", "lds_bank_conflict_per_wave_cycle"]
    out = [hdr]
    avg = lambda v: sum(v) / len(v) if v else 0.0
    for k, c in sorted(byk.items()):
        wc = avg(c["SQ_WAVE_CYCLES"])
        gui = avg(c["GRBM_GUI_ACTIVE"]) / 8.0
        out.append([k, len(c["FETCH_SIZE"]) or len(c["SQ_WAVE_CYCLES"]), "%.0f" % avg(c["FETCH_SIZE"]), "%.0f" % avg(c["WRITE_SIZE"]),
                    "%.1f" % ((2 * avg(c["FETCH_SIZE"]) + avg(c["WRITE_SIZE"])) * 1024 / 1e6),
                    "%.3f" % (avg(c["SQ_WAIT_ANY"]) / wc) if wc else "", "%.3f" % (avg(c["SQ_WAIT_INST_ANY"]) / wc) if wc else "",
                    "%.3f" % (avg(c["SQ_ACTIVE_INST_ANY"]) / wc) if wc else "",
                    "%.3f" % (avg(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui * 1024)) if gui else "",
                    "%.4f" % (avg(c["SQ_LDS_BANK_CONFLICT"]) / wc) if wc else ""])
    with open(os.path.join(dst, "pmc_by_kernel.csv"), "w", newline="") as f:
        csv.writer(f).writerows(out)
    if n_patch:
        tj = os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_traffic.json")
        try:
            d = json.load(open(tj))
        except (OSError, ValueError):
            d = {}
        build = None
        try:
            build = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]).get("build")
        except (OSError, ValueError, IndexError):
            pass
        d["conv3x3_patch_mfma|320|bf16|32"] = {"hbm_bytes_per_launch": int(patch_hbm * 1e6 / n_patch), "launches": n_patch, "build": build,
                                               "source": os.path.basename(os.path.abspath(dst)) + "/per_layer.csv (2 x FETCH_SIZE + WRITE_SIZE)"}
        json.dump(d, open(tj, "w"), indent=1)
        print("conv3x3_patch family in the timed loop (kernel trace): %.1f us over %d launches = %.1f TFLOP/s = %.3f of peak; HBM %.1f MB per launch"
              % (patch_trace_us, n_patch, patch_gflop / patch_trace_us * 1e3, patch_gflop / patch_trace_us * 1e3 / PEAK, patch_hbm / n_patch))
        if graph_line:
            gus, gn = graph_line
            print("conv3x3 family in the hipGraph REPLAY (kernel trace of what bench.py times): %.1f us per launch over %d launches = %.1f TFLOP/s = %.3f of peak"
                  % (gus, gn, patch_gflop / n_patch / gus * 1e3, patch_gflop / n_patch / gus * 1e3 / PEAK))
            with open(os.path.join(dst, "graph_replay_family.txt"), "w") as f:
                f.write("conv3x3 family (conv3x3_patch_kernel + conv3x3_pp_kernel + conv3x3_ws_kernel), rocprofv3 --kernel-trace of `bench.py --graph 1`: %.2f us per launch over %d launches "
                        "= %.1f TFLOP/s = %.4f of 2500\n" % (gus, gn, patch_gflop / n_patch / gus * 1e3, patch_gflop / n_patch / gus * 1e3 / PEAK))


if __name__ == "__main__":
    main()
