#!/usr/bin/env python3
"""bench.py -- frames/sec of the dual-refinement detector hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- build_net(...)'s forward + Detect -- over one batch of
synthetic frames already resident in HBM (BASELINE.json configs[1]: dualrefinedet_vggbn 320x320,
bf16, batch 32 per GPU, multihead, synthetic VOC-shaped frames, synthetic weights).  N>1: every rank
runs its own batch (frames shard, weights arrive by one RCCL broadcast, no per-frame collective):
weak scaling, value = all ranks' frames / max-over-ranks time.

The JSON line also carries
  roofline     : the dominant kernel family (conv3x3_patch_mfma): algorithmic FLOPs of its launches in one
                 step / their summed hipEvent durations IN THE PRODUCTION SCHEDULE (events recorded by the library on the
                 stream each launch runs on, side lanes on), against the dense bf16 MFMA peak; `single_stream` gives the
                 same with every launch alone on one stream; `traffic` = HBM-side bytes per launch from the committed
                 rocprofv3 PMC passes (profiles/pmc_traffic.json)
  parity       : the other half of BASELINE's metric ("box L-inf vs CPU ref"): decoded boxes and softmax scores of
                 the TIMED dtype (and of the fp32 and fp16 modes beside it) against the fp32 CPU oracle on one frame
  modes        : the OTHER precisions of the same workload, each timed in this run as hipGraph replays (frames/s,
                 ms_per_step, dominant-kernel fraction of ITS OWN peak: fp32 against the 157.3 TFLOP/s fp32 matrix peak)
                 -- the fp32 mode is the one that meets north_star's "within 1e-3" clause
  detections   : per precision, what the error does to the FINAL detections of >= 8 frames: (class, slot) rows of
                 Detect's output against the oracle's Detect on the fp32 oracle outputs (identical occupancy, matched at
                 IoU >= 0.9, moved > 1 px, appeared, vanished)
  cpu_baseline : the CPU oracle (torch-CPU convs + C deformable conv + C Detect) on a bounded sample
                 of the same workload on this host's cores (rank 0, N=1 only)

`--config N` selects another BASELINE.json configuration (default 2, the headline -- the driver's command is unchanged):
  3  dualrefinedet_vggbn 512x512 fp16 batch 16 (same line shape, MFMA roofline)
  4  dualrefinedet_mobilenet 320x320 batch 64 per GPU: HBM roofline (the MobileNet models are bandwidth-bound) --
     algorithmic bytes of a step (layer-boundary activations once in / once out + the weights once) / the forward's time,
     against 8 TB/s; the per-family table (`kernels`) carries each family's algorithmic bytes and time
  5  TRN clips (evaluate_trn.py:438-467): 8 clips x 4 frames per step, interval 4: per clip 1 static forward (ssd4scale_vgg,
     loc maps out) + 4 temporal forwards (ssd4scale_vgg deform=True, 8 deformable groups; offsets from the key frame) + 4
     Detect calls; value = frames/s (clips/s beside it), MFMA roofline of the 3x3 conv family over both nets
--size / --dtype / --batch override the preset.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own N ranks (fresh child processes, one per
GPU, RCCL over 127.0.0.1) BEFORE anything in this process touches a GPU and relays rank 0's JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md, dense
GFLOP_PER_FRAME = {320: 77.466, 512: 198.314}                   # BASELINE.md section 2 (multihead)
HBM_PEAK_GBS = 8000.0                                            # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# BASELINE.json configs[1..4] (configs[0] is the reference's own CPU plumbing case)
CONFIGS = {
    2: dict(model="dualrefinedet_vggbn", size=320, dtype="bf16", batch=32),
    3: dict(model="dualrefinedet_vggbn", size=512, dtype="fp16", batch=16),
    # (config 4 names no dtype: fp16 is the deployment default for such configs -- tests/test_gpu_net.py -- and bf16's box error on this
    # model, 0.54 of the frame at a sampling discontinuity, is not something to ship; `--dtype bf16` times the other one)
    4: dict(model="dualrefinedet_mobilenet", size=320, dtype="fp16", batch=64),
    5: dict(model="trn_ssd4scale_vgg", size=320, dtype="bf16", batch=8),      # batch = clips of 4 frames
}
# algorithmic GFLOP per frame and layer-boundary activation elements per frame (M)
MOBILENET_GFLOP, MOBILENET_ACT_MELEMS = 22.340, 55.2
NCLS = 21                                                        # --classes: VOC 21 (BASELINE's configs), VID 31 (evaluate_trn.py:526), COCO 81 (evaluate_coco.py:245)


def head_gflop_delta(size, multihead=True, deform_taps=34, heads_per_level=1):
    """algorithmic GFLOP per frame that the conf heads add per class beyond 21: 3 anchors x taps x 256 channels x 2 over the four
    pyramid levels ((size/8)^2 (1 + 1/4 + 1/16 + 1/64) pixels); BASELINE.md's figures are quoted at 21 classes"""
    px = (size / 8.0) ** 2 * (1 + 0.25 + 0.0625 + 0.015625)
    return 2.0 * px * 3 * deform_taps * 256 * (NCLS - 21) / 1e9
SSD4SCALE_VGG_GFLOP = 65.442


def pmc_traffic(kernel, args, build):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    FETCH_SIZE doubled per the gfx950 correction for 16-B/lane streams + WRITE_SIZE, both x1024).  It is a figure
    FROM THE PROFILES, not a measurement of this run: (value, source) -- value is None when no pass exists for this
    exact workload or when the passes were taken on another build of the library."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
        e = d.get("%s|%d|%s|%d" % (kernel, args.size, args.dtype, args.batch))
        if not e:
            return None, "no PMC pass committed for this workload"
        if e.get("build") and e["build"] != build:
            return None, "profiles/pmc_traffic.json is of build %r, this run is %r" % (e["build"], build)
        return e["hbm_bytes_per_launch"], "profiles/pmc_traffic.json (rocprofv3 PMC passes, %s): not measured in this run" % e.get("build", "build not recorded")
    except (OSError, ValueError, KeyError):
        return None, "profiles/pmc_traffic.json unreadable"


NEAR_EPS = {"fp32": 1e-4, "bf16": 0.03, "fp16": 0.004}   # pixels with a deformable tap this close to a sampling discontinuity


def parity_vs_oracle(size, dtypes, dev):
    """Box / score error of the HIP path in each of `dtypes` against the fp32 CPU oracle on one synthetic frame.
    Boxes = two-stage decode (layers/box_utils.py:176-195, what Detect does first) of (arm_loc, odm_loc) in normalised
    image coordinates.  The deformable sampling rule is discontinuous at the map border
    (deform_conv_cuda_kernel.cu:195), so prior rows of pixels with a tap within NEAR_EPS of it flip by O(1) in ANY
    implementation that rounds differently; they are counted and left out of the L-inf / p99.9 (not of the mean)."""
    import numpy as np
    import torch
    from oracle import net_ref
    from oracle import oracle as orc
    from tdrn_amd.layers.box_utils import center_size, decode
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    from tdrn_amd.utils import synth
    net = build_net("test", size, NCLS, 1024, 1, True, True)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    x = synth.synth_frames(1, size, seed=5)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    taps = {}
    r_arm, _, r_odm, r_conf = net_ref.drn_vggbn_forward(sd, x, NCLS, True, True, taps=taps)
    cfg = dict(feature_maps=[size // 8, size // 16, size // 32, size // 64], min_dim=size, steps=[8, 16, 32, 64],
               min_sizes=[32, 64, 128, 256], max_sizes=[], aspect_ratios=[[2]] * 4, variance=[0.1, 0.2], clip=True,
               flip=True, name="bench")
    pri = orc.prior_box(cfg)
    r_box = orc.decode(r_odm.numpy()[0], orc.center_size(orc.decode(r_arm.numpy()[0], pri)))
    out = {"frames": 1, "reference": "fp32 CPU oracle (oracle/net_ref.py + oracle/tdrn_oracle.c)", "box_unit": "normalised image coordinates"}
    pri_d = torch.from_numpy(pri).to(dev)
    for dt in dtypes:
        net.set_compute_dtype(dt)
        arm, _, odm, conf = net(torch.from_numpy(x).to(dev))
        box = decode(odm[0], center_size(decode(arm[0], pri_d, [0.1, 0.2])), [0.1, 0.2]).cpu().numpy()   # tdrn_decode / tdrn_center_size
        near = net_ref.border_rows(taps, True, NEAR_EPS[dt])
        eb = np.abs(box - r_box)
        es = np.abs(conf.cpu().numpy() - r_conf.numpy().reshape(conf.shape))
        out[dt] = {"box_linf": float(eb[~near].max()), "box_p999": float(np.quantile(eb[~near], 0.999)), "box_mean": float(eb.mean()),
                   "score_linf": float(es[~near].max()), "score_mean": float(es.mean()),
                   "rows_near_discontinuity": int(near.sum()), "box_linf_all_rows": float(eb.max())}
    return out


def cpu_baseline(size, frames=2):
    """Oracle forward + Detect on `frames` frames; returns the cpu_baseline object."""
    import torch
    from oracle import net_ref
    from oracle import oracle as orc
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    from tdrn_amd.utils import synth
    net = build_net("test", size, NCLS, 1024, 1, True, True)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    # torch's CPU convs peak at ~16 threads on the 256-core GPU host (measured: 8/16/32/64/128 threads
    # -> 0.066/0.046/0.065/0.20/0.40 s per trunk); all-core runs are 100x slower from oversubscription.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = dict(feature_maps=[size // 8, size // 16, size // 32, size // 64], min_dim=size, steps=[8, 16, 32, 64],
               min_sizes=[32, 64, 128, 256], max_sizes=[], aspect_ratios=[[2]] * 4, variance=[0.1, 0.2], clip=True,
               flip=True, name="bench")
    pri = orc.prior_box(cfg)
    x = synth.synth_frames(frames, size, seed=0)
    net_ref.drn_vggbn_forward(sdt, x[:1], NCLS, True, True)          # warm-up (thread pools, page-in)
    t0 = time.perf_counter()
    for i in range(frames):
        arm, _, odm, conf = net_ref.drn_vggbn_forward(sdt, x[i:i + 1], NCLS, True, True)
        orc.detect(odm.numpy(), conf.numpy(), pri, arm.numpy(), (500, 375, 500, 375), num_classes=NCLS)
    dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same workload, batch 1, fp32 (oracle/: torch-CPU convs + C deformable "
                      "conv + C Detect), %.1f s" % (frames, dt)}


# Engines, captured graphs and the streamed pipeline are kept until the process ends: on ROCm 7.2 hipGraphLaunch crashed (SIGSEGV)
# for a graph captured AFTER earlier graphs / engines of the process had been destroyed (seen with the fp32 plan after two such
# destructions, in either order of the blocks below); nothing is freed, nothing crashes -- 288 GB of HBM holds three engines.
KEEP_ALIVE = []


def detection_agreement(size, dtypes, dev, frames=8):
    """What each precision's error does to the FINAL detections (layers/functions/detection.py:25-70): Detect (HIP) on the
    HIP net's outputs against the oracle's Detect on the fp32 oracle's outputs, over `frames` synthetic frames, boxes in
    the 500 x 375 pixel units evaluate.py uses.  Per dtype: rows (= occupied (image, class, slot) entries) on each side,
    rows whose (class, slot) is occupied on both sides, greedy one-to-one matches at IoU >= 0.9 inside each (image, class),
    matched rows that moved by more than 1 px in any coordinate, rows that appeared / vanished."""
    import numpy as np
    import torch
    from oracle import net_ref
    from oracle import oracle as orc
    from tdrn_amd.layers import Detect
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    from tdrn_amd.utils import synth
    net = build_net("test", size, NCLS, 1024, 1, True, True)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x = synth.synth_frames(frames, size, seed=77)
    cfg = dict(feature_maps=[size // 8, size // 16, size // 32, size // 64], min_dim=size, steps=[8, 16, 32, 64],
               min_sizes=[32, 64, 128, 256], max_sizes=[], aspect_ratios=[[2]] * 4, variance=[0.1, 0.2], clip=True,
               flip=True, name="bench")
    pri = orc.prior_box(cfg)
    scale = (500.0, 375.0, 500.0, 375.0)
    ref = []
    for i in range(frames):
        arm, _, odm, conf = net_ref.drn_vggbn_forward(sd, x[i:i + 1], NCLS, True, True)
        ref.append(orc.detect(odm.numpy(), conf.numpy(), pri, arm.numpy(), scale, num_classes=NCLS)[0])     # (C, top_k, 5)
    pri_d = torch.from_numpy(pri).to(dev)

    def iou(a, b):
        iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
        ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
        inter = iw * ih
        aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-12)
    out = {"frames": frames, "box_unit": "pixels of a 500 x 375 frame", "reference": "oracle Detect on the fp32 CPU oracle's outputs"}
    for dt in dtypes:
        net.set_compute_dtype(dt)
        r = net.engine(dev).forward(torch.from_numpy(x).to(dev))
        got = Detect(NCLS, 0, 200, 0.01, 0.45).forward(r["odm_loc"], r["conf"], pri_d, arm_loc_data=r["arm_loc"],
                                                       scale=torch.tensor(scale)).cpu().numpy()
        n_ref = n_got = same_slot = matched = moved = 0
        for i in range(frames):
            for c in range(1, NCLS):
                a, b = ref[i][c], got[i][c]
                ka, kb = a[:, 0] > 0, b[:, 0] > 0
                n_ref += int(ka.sum()); n_got += int(kb.sum()); same_slot += int((ka & kb).sum())
                if not ka.any() or not kb.any():
                    continue
                ba, bb = a[ka][:, 1:], b[kb][:, 1:]
                m = iou(ba, bb)
                used = np.zeros(len(bb), bool)
                for j in range(len(ba)):                      # oracle rows in score order take their best free partner
                    k = int(np.argmax(np.where(used, -1.0, m[j])))
                    if not used[k] and m[j, k] >= 0.9:
                        used[k] = True
                        matched += 1
                        moved += int(np.abs(ba[j] - bb[k]).max() > 1.0)
        out[dt] = {"rows_oracle": n_ref, "rows_hip": n_got, "rows_same_class_slot": same_slot, "matched_iou_0.9": matched,
                   "matched_moved_gt_1px": moved, "vanished": n_ref - matched, "appeared": n_got - matched}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh ranks (one per GPU) before this process
    has touched a GPU, relay rank 0's stdout (the one JSON line), exit non-zero if any rank does."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    one_device = os.environ.get("TDRN_DIST_ONE_DEVICE") == "1"      # test-only: every rank on device 0 (needs TDRN_DIST_BACKEND=gloo)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_device else str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: ranks failed: %r" % bad, file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def streamed_block(engines, launch_mode, NF, NSL, B, S_note, args, fps, world, tdist, torch, dev, pri, with_note):
    """The streamed mode of a configuration (uint8 frames from pinned host memory in, detections out; tdrn_amd/stream.py): builds the
    FrameStream the way the headline was launched and times args.steps steps max(3, reps / 2) times.  With two eager pipelines the
    copies can travel on the pipelines' own streams ("own") or on a calibrated pair of copy streams ("stream"); which is faster depends
    on how the run's streams alias onto the hardware queues (round 5: own; round 6's boxes: stream, 0.93 against 0.88 of the resident
    rate), so both are built, timed over a few steps, and the faster one is measured -- the line says which (`copies`,
    `copy_placement_frames_per_s`).  TDRN_STREAM_COPY_IN forces one."""
    import numpy as np
    from tdrn_amd.layers import Detect
    from tdrn_amd.stream import FrameStream
    forced = os.environ.get("TDRN_STREAM_COPY_IN")
    cands = [forced] if forced else (["stream", "own"] if (launch_mode == "eager" and NF == 2) else ["stream"])
    rng = np.random.RandomState(7)
    frames = [torch.from_numpy(rng.randint(0, 256, size=(B, 375, 500, 3), dtype=np.uint8)) for _ in range(NSL)]
    built, table = {}, {}
    for c in cands:
        fs = FrameStream(list(engines) if NF > 1 else engines[0], Detect(NCLS, 0, 200, 0.01, 0.45), pri, B, slots=NSL, graph=launch_mode != "eager", copy_in=c)
        for sl in range(NSL):
            fs.pinned_in(sl).copy_(frames[sl])
        fs.prime()
        for k in range(2 * NSL):
            fs.run()
        fs.drain()
        built[c] = fs
        if len(cands) > 1:
            n = max(8, min(args.steps, 24))
            best = 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(n):
                    fs.run()
                fs.drain()
                best = min(best, (time.perf_counter() - t0) / n)
            table[c] = round(world * B / best, 2)
    pick = max(table, key=table.get) if table else cands[0]
    fs = built[pick]
    KEEP_ALIVE.extend(built.values())
    t_stream = []
    for _ in range(max(3, args.reps // 2)):
        tdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            fs.run()
        fs.drain()
        tdist.barrier()
        t_stream.append(tdist.max_over_ranks(time.perf_counter() - t0, dev))
    t_stream.sort()
    ts = t_stream[len(t_stream) // 2]
    fs.prime()
    got = fs.result(fs.run()).clone()
    want = fs.eager(fs.pinned_in(0).to(dev)).cpu()
    blk = {"frames_per_s": round(world * B * args.steps / ts, 2), "ms_per_step": round(ts / args.steps * 1e3, 4),
           "vs_resident": round((world * B * args.steps / ts) / fps, 4), "repetitions": len(t_stream),
           "copy_streams_picked": (fs.calibration or {}).get("picked"), "pipeline_streams_picked": (fs.pipeline_calibration or {}).get("picked"),
           "launch": "hipGraph replay" if fs.graph else "eager",
           "copies": "on the pipelines' own streams" if fs.copy_in == "own" else ("copy-in / copy-out streams of high priority (no calibration)" if getattr(fs, "copy_priority", False) else "copy-in / copy-out streams (picked by calibration)"),
           "copy_placement_frames_per_s": table or None,
           "detections_identical_to_unstreamed": bool(torch.equal(got, want)),
           "per_step": "copy-in: H2D %.1f MB of uint8 BGR 500x375 frames (pinned), two batches ahead | compute streams (slot s on pipeline s mod in-flight): per slot (one hipGraph, or the same launches issued eagerly: `launch`) tdrn_preprocess_u8 (resize, frames stay uint8), net (mean subtracted in the first conv's loader), Detect | copy-out: D2H %.1f MB of detections; %d slots, event-chained"
                       % (B * 375 * 500 * 3 / 1e6, got.numel() * 4 / 1e6, NSL)}
    if with_note:
        blk["note"] = "the resident figure (`value`) starts from fp32 frames already preprocessed in HBM; this one includes the resize kernel and both copies"
    return blk


NUMA_PIN = None


def pin_rank_to_numa_node():
    """N > 1: every rank pins itself to the CPUs of its GPU's NUMA node BEFORE its first GPU call (tdrn_amd/dist.py, sysfs only):
    the frame feeders of 8 ranks otherwise share whatever cores the launcher left them on."""
    global NUMA_PIN
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and os.environ.get("TDRN_NUMA_PIN", "1") != "0":
        from tdrn_amd import dist as tdist
        NUMA_PIN = tdist.pin_to_gpu_numa_node(int(os.environ.get("LOCAL_RANK", "0")), world)


def _timed(stepper, steps, reps, tdist, torch, dev):
    out = []
    for _ in range(reps):
        tdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            stepper(k)
        torch.cuda.synchronize()
        tdist.barrier()
        out.append(tdist.max_over_ranks(time.perf_counter() - t0, dev))
    return out


def _choose_launch(graph_flight, make_eager, steps, frames, tdist, torch, dev):
    """Two ways to put the same two pipelines in flight: one hipGraph replay per step, or the step's launches issued eagerly from
    this thread (the reference's own drivers launch eagerly).  On ROCm 7.2 two replays on two streams hardly overlap
    (scripts/dev/inflight_timeline.py: step k + 1 starts when step k ends) while eager pipelines do -- if their main lanes sit on
    different hardware queues, which InFlight.pick_streams settles by a short calibration.  Which way is faster depends on the host
    (it has to keep up with ~70 launches per step), so both are timed here over the same steps and the timed region uses the faster;
    the line says which (`config.launch`, `launch_calibration_frames_per_s`).  Returns (flight, mode, calibration)."""
    eager = make_eager()
    for k in range(4):
        eager.launch(k)                                  # (lanes, workspaces, LDS attributes of the clone exist before the calibration)
    eager.pick_streams()
    cal, n = {}, max(8, min(steps, 40))
    for name_, fl_ in (("hipGraph replay", graph_flight), ("eager", eager), ("hipGraph replay", graph_flight), ("eager", eager)):
        for k in range(4):
            fl_.launch(k)
        cal[name_] = min(cal.get(name_, 1e9), min(_timed(fl_.launch, n, 1, tdist, torch, dev)) / n)
    table = {k_: round(frames / v_, 2) for k_, v_ in cal.items()}
    if cal["eager"] < cal["hipGraph replay"] * 0.995:
        return eager, "eager", table
    return graph_flight, "hipGraph replay", table


def _family_table(stats_prod, stats_solo, mult=1):
    solo_ms = {s["name"]: s["ms"] for s in stats_solo}
    return {s["name"]: {"ms": round(s["ms"] * mult, 4), "ms_single_stream": round(solo_ms.get(s["name"], 0.0) * mult, 4), "launches": s["launches"] * mult,
                        "gflop": round(s["flops"] * mult / 1e9, 3), "gbyte": round(s["bytes"] * mult / 1e9, 4),
                        "gbyte_per_s": round(s["bytes"] / 1e9 / (s["ms"] * 1e-3), 1) if s["ms"] > 0 else 0.0,
                        "tflop_per_s": round(s["flops"] / 1e12 / (s["ms"] * 1e-3), 1) if s["ms"] > 0 else 0.0} for s in stats_prod}


def main_other(args):
    """BASELINE configs #4 (dualrefinedet_mobilenet 320, batch 64, HBM roofline) and #5 (TRN clips, MFMA roofline): the same
    contract and line shape as the headline configuration."""
    import numpy as np
    import torch
    from tdrn_amd import _lib
    from tdrn_amd import dist as tdist
    from tdrn_amd.data import mb_cfg
    from tdrn_amd.engine import GraphedCall
    from tdrn_amd.layers import Detect, PriorBox
    from tdrn_amd.utils import synth

    rank, local_rank, world = tdist.init()
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    build = _lib.lib().tdrn_version().decode()
    S, B, NB = args.size, args.batch, max(1, args.batches)
    es = 4 if args.dtype == "fp32" else 2
    pri = PriorBox(mb_cfg["VOC_320" if S == 320 else "VOC_512_RefineDet"]).forward().to(dev)
    scale = [500.0, 375.0, 500.0, 375.0]
    trn = args.config == 5

    def make(modname, a, kw, seed, plan_flags=0):
        import importlib
        net = importlib.import_module("tdrn_amd.model." + modname).build_net("test", *a, **kw)
        if plan_flags:
            net.set_plan_flags(plan_flags)
        net.set_compute_dtype(args.dtype)
        sd = None
        if rank == 0:
            sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        t0 = time.perf_counter()
        eng = net.adopt_broadcast_weights(src=0, device=dev) if tdist.active() else net.engine(dev)
        torch.cuda.synchronize()
        return net, sd, eng, (time.perf_counter() - t0) * 1e3

    if trn:
        # (overlapped mode: the static net runs on a second stream of the step's capture, and its plan uses that one stream only -- a
        # forward that forks its own stream lanes from a stream which itself joined the capture by an event takes
        # hipStreamEndCapture down on ROCm 7.2, scripts/dev/trn_overlap_probe.py)
        overlap = args.trn_mode == "batched" and bool(args.trn_overlap)
        stat, sd_s, eng_s, bc1 = make("ssd4scale_vgg", (S, NCLS), dict(c7_channel=1024, bn=False, deform=False), 0,
                                      _lib.PLAN_ONE_STREAM if overlap else 0)
        temp, sd_t, eng_t, bc2 = make("ssd4scale_vgg", (S, NCLS), dict(c7_channel=1024, bn=False, deform=True), 1)
        bcast_ms = bc1 + bc2
        FPC = 4                                                  # frames per clip = the key-frame interval
        det = [Detect(NCLS, 0, 200, 0.01, 0.45) for _ in range(FPC)]
        # a step's clips FRAME-major: (FPC, B, 3, S, S) -- [0] = the B key frames, .view(FPC * B, ...) = all frames, frame i of clip i % B
        xb = [torch.from_numpy(synth.synth_frames(B * FPC, S, seed=100 + rank + 1000 * j)).to(dev).view(FPC, B, 3, S, S) for j in range(NB)]
        batched = args.trn_mode == "batched"
        trn_side, trn_ev = torch.cuda.Stream(dev), torch.cuda.Event()
        KEEP_ALIVE.append((trn_side, trn_ev))

        def one_step(clips, batched=batched, stat=stat, temp=temp, det=det, trn_side=trn_side, trn_ev=trn_ev):
            # evaluate_trn.py:438-467 over B clips at once: key frame -> static net (anchors + loc maps) -> temporal net (offsets
            # from the key frame's loc maps, reused by the frames up to the next key frame); Detect on the static anchors.
            if batched and args.trn_overlap:
                # ... and the static net's forward (8 key frames: small launches that leave CUs idle) runs on a second stream BESIDE
                # the temporal net's trunk, which does not depend on it; the temporal forward waits for the static net's event right
                # before its first read of the loc maps (tdrn_net_io.reserved[2])
                main = torch.cuda.current_stream(dev)
                trn_side.wait_stream(main)
                with torch.cuda.stream(trn_side):
                    s_loc, _, maps = stat(clips[0], ret_loc=True)
                    trn_ev.record(trn_side)     # (one event object for the process: destroyed inside a stream capture it takes hipStreamEndCapture down)
                loc, conf = temp(clips.view(FPC * B, 3, S, S), ref_loc=maps, ref_event=trn_ev)[:2]
                main.wait_stream(trn_side)
                for t_ in [s_loc] + list(maps):
                    t_.record_stream(main)
                return conf if args.no_detect else det[0].forward(loc, conf, pri, arm_loc_data=s_loc.repeat(FPC, 1, 1), scale=scale)
            s_loc, _, maps = stat(clips[0], ret_loc=True)
            if batched:
                # the frames of an interval depend on the key frame only through its offsets: ONE temporal forward over all
                # FPC * B frames (tdrn_net_io.reserved[1]: frame i reads the offsets of key frame i % B) and one Detect call
                loc, conf = temp(clips.view(FPC * B, 3, S, S), ref_loc=maps)[:2]
                return conf if args.no_detect else det[0].forward(loc, conf, pri, arm_loc_data=s_loc.repeat(FPC, 1, 1), scale=scale)
            outs, offs = [], None
            for f in range(FPC):                       # the reference's order: frame by frame, offset_list cached from the key frame
                if f == 0:
                    loc, conf, offs = temp(clips[f], ref_loc=maps, ret_off=True)
                else:
                    loc, conf = temp(clips[f], offset_list=offs)
                outs.append(conf if args.no_detect else det[f].forward(loc, conf, pri, arm_loc_data=s_loc, scale=scale))
            return outs
        frames_per_step = B * FPC
        gflop_step = B * (SSD4SCALE_VGG_GFLOP + FPC * SSD4SCALE_VGG_GFLOP)     # (plain heads ~ deformable heads in FLOPs: same taps, same channels)
        engines = [(eng_s, 1), (eng_t, 1 if batched else FPC)]
        name = "TRN ssd4scale_vgg static + temporal (deform, 8 groups)"
    else:
        net, sd, eng, bcast_ms = make("dualrefinedet_mobilenet", (S, NCLS), dict(def_groups=1, multihead=True), 0)
        det = Detect(NCLS, 0, 200, 0.01, 0.45)
        xb = [torch.from_numpy(synth.synth_frames(B, S, seed=100 + rank + 1000 * j)).to(dev) for j in range(NB)]

        def one_step(x):
            r = eng.forward(x)
            return r["conf"] if args.no_detect else det.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
        frames_per_step = B
        gflop_step = B * MOBILENET_GFLOP
        engines = [(eng, 1)]
        name = "dualrefinedet_mobilenet multihead"

    NF = 1
    launch_mode, launch_cal = ("hipGraph replay" if args.graph else "eager"), None
    trn_flight = None
    if trn and args.graph and batched and args.trn_overlap and args.in_flight > 1 and args.launch == "auto" and NB % 2 == 0:
        # config 5 with two steps in flight: a second pipeline = twins of the static and the temporal model (own engines over the same
        # weight blobs, own Detect, side stream and event), the two launched eagerly on calibrated streams; timed against the
        # one-step-at-a-time hipGraph replays below and used if faster (as the headline configuration does)
        from tdrn_amd.engine import InFlight
        stat2, temp2 = stat.pipeline_twin(dev), temp.pipeline_twin(dev)
        det2 = [Detect(NCLS, 0, 200, 0.01, 0.45) for _ in range(FPC)]
        side2, ev2 = torch.cuda.Stream(dev), torch.cuda.Event()
        KEEP_ALIVE.append((stat2, temp2, det2, side2, ev2))
        step2 = lambda clips: one_step(clips, stat=stat2, temp=temp2, det=det2, trn_side=side2, trn_ev=ev2)

        def make_trn_flight():
            fl_ = InFlight(None, eng_t, xb, graph=False, steps=[lambda clips: one_step(clips), step2],
                           engines=[eng_s, eng_t, stat2.engine(dev), temp2.engine(dev)])
            KEEP_ALIVE.append(fl_)
            return fl_
        trn_flight = make_trn_flight
    if args.graph and not trn and args.in_flight > 1:
        # config 4: two steps in flight, as the headline configuration (tdrn_amd.engine.InFlight)
        from tdrn_amd.engine import InFlight
        NF = args.in_flight
        while NB % NF:
            NF -= 1

        def make_step(e):
            d = Detect(NCLS, 0, 200, 0.01, 0.45)

            def step_of(x):
                r = e.forward(x)
                return r["conf"] if args.no_detect else d.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
            return step_of
        flight = InFlight(make_step, eng, xb, n=NF, graph=True)
        KEEP_ALIVE.append(flight)
        if args.launch == "auto":
            def make_eager():
                fl_ = InFlight(make_step, eng, xb, n=NF, graph=False)
                KEEP_ALIVE.append(fl_)
                return fl_
            flight, launch_mode, launch_cal = _choose_launch(flight, make_eager, args.steps, world * frames_per_step, tdist, torch, dev)
        step = flight.launch
        engines = [(e_, 1) for e_ in flight.engines]
    elif args.graph:
        graphs = [GraphedCall(one_step, xb[j]) for j in range(NB)]
        KEEP_ALIVE.append(graphs)
        step = lambda k: graphs[k % NB](graphs[k % NB].inputs[0])
        if trn_flight is not None:
            class _Replays(object):                      # (the one-step-at-a-time replays behind InFlight's launch(k))
                launch = staticmethod(step)
            chosen, launch_mode, launch_cal = _choose_launch(_Replays, trn_flight, args.steps, world * frames_per_step, tdist, torch, dev)
            if launch_mode == "eager":
                NF, step = 2, chosen.launch
                engines = [(eng_s, 1), (eng_t, 1)]
    else:
        step = lambda k: one_step(xb[k % NB])
    for k in range(args.warmup):
        step(k)
    reps = sorted(_timed(step, args.steps, max(1, args.reps), tdist, torch, dev))
    for e_, _m in engines:
        e_.check()                  # tdrn_net_check: a device-side hand-off that timed out inside the replays above fails the run HERE
    dt = reps[len(reps) // 2]
    fps = world * frames_per_step * args.steps / dt
    one_at_a_time = None
    if NF > 1:
        if not trn:
            engines = engines[:1]                   # (the accounting passes below run engine 0 alone)
        g1 = graphs if trn else [GraphedCall(one_step, xb[j]) for j in range(NB)]
        KEEP_ALIVE.append(g1)
        st1 = lambda k: g1[k % NB](g1[k % NB].inputs[0])
        for k in range(3):
            st1(k)
        r1 = sorted(_timed(st1, args.steps, 3, tdist, torch, dev))
        one_at_a_time = {"frames_per_s": round(world * frames_per_step * args.steps / r1[1], 2), "ms_per_step": round(r1[1] / args.steps * 1e3, 4), "repetitions": 3}
    # config 5, batched mode: the same clips in the reference loop's order (one temporal forward and one Detect per frame index),
    # timed in the same process -- what the batching is worth
    frame_loop = None
    if trn and batched and not args.no_frame_loop:
        loop_step = lambda clips: one_step(clips, batched=False)
        if args.graph:
            lg = [GraphedCall(loop_step, xb[j]) for j in range(NB)]
            KEEP_ALIVE.append(lg)
            lstep = lambda k: lg[k % NB](lg[k % NB].inputs[0])
        else:
            lstep = lambda k: loop_step(xb[k % NB])
        for k in range(args.warmup):
            lstep(k)
        lr = sorted(_timed(lstep, args.steps, max(1, min(3, args.reps)), tdist, torch, dev))
        ldt = lr[len(lr) // 2]
        frame_loop = {"frames_per_s": round(world * frames_per_step * args.steps / ldt, 2), "ms_per_step": round(ldt / args.steps * 1e3, 4),
                      "what": "1 static + 4 temporal forwards of %d frames + 4 Detect calls per step (--trn-mode frames)%s" % (
                          B, "; the static net here is the ONE-STREAM plan the overlapped schedule needs (its multi-lane plan is ~3 %% faster alone: this "
                             "baseline is slightly pessimistic)" if overlap else "")}

    # forward-only time of a step (eager, no Detect), and the per-family accounting of one step
    def fwd_only():
        if trn:
            clips = xb[0]
            _, _, maps = stat(clips[0], ret_loc=True)
            if batched:
                temp(clips.view(FPC * B, 3, S, S), ref_loc=maps)
                return
            offs = None
            for f in range(FPC):
                if f == 0:
                    _, _, offs = temp(clips[f], ref_loc=maps, ret_off=True)
                else:
                    temp(clips[f], offset_list=offs)
        else:
            eng.forward(xb[0])
    fwd_only()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        fwd_only()
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t1) / 5 * 1e3

    def profiled(mode):
        fam = {}
        for e, mult in engines:
            e.set_profile(mode)
        fwd_only(); fwd_only()
        torch.cuda.synchronize()
        per_engine = []
        for e, mult in engines:
            st = e.kernel_stats()
            per_engine.append((st, e.op_stats() if args.per_op else None, mult))
            for srow in st:
                f = fam.setdefault(srow["name"], dict(name=srow["name"], launches=0, flops=0.0, bytes=0.0, ms=0.0))
                for kk in ("launches", "flops", "bytes", "ms"):
                    f[kk] += srow[kk] * mult
            e.set_profile(0)
        return list(fam.values()), per_engine
    solo, per_solo = profiled(1)
    prod, per_prod = profiled(2)
    if args.per_op and rank == 0:
        for (st, ops, mult), (st2, ops2, _) in zip(per_solo, per_prod):
            p2 = {o["name"]: o["ms"] for o in ops2}
            print("%-44s %11s %11s %8s %9s %8s %9s   (x%d per step)" % ("launch", "alone us", "in step us", "GFLOP", "TFLOP/s", "GB", "GB/s", mult), file=sys.stderr)
            for o in ops:
                tf = o["flops"] / (o["ms"] * 1e-3) / 1e12 if o["ms"] > 0 else 0.0
                gb = o["bytes"] / (o["ms"] * 1e-3) / 1e9 if o["ms"] > 0 else 0.0
                print("%-44s %11.1f %11.1f %8.1f %9.1f %8.3f %9.0f" % (o["name"], o["ms"] * 1e3, p2.get(o["name"], 0.0) * 1e3, o["flops"] / 1e9, tf, o["bytes"] / 1e9, gb), file=sys.stderr)
    kernels = _family_table(prod, solo)
    dom = max(prod, key=lambda f: f["ms"])
    dom_solo = next(f for f in solo if f["name"] == dom["name"])
    if trn:
        peak = PEAK_TFLOPS[args.dtype]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        ach_s = dom_solo["flops"] / (dom_solo["ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": None, "traffic_source": "no PMC pass committed for this workload", "launches_per_step": dom["launches"],
                    "gflop_per_launch": round(dom["flops"] / 1e9 / dom["launches"], 2), "us_per_launch": round(dom["ms"] * 1e3 / dom["launches"], 2),
                    "mode": "production schedule (side lanes on)",
                    "single_stream": {"achieved": round(ach_s, 2), "frac": round(ach_s / peak, 4), "us_per_launch": round(dom_solo["ms"] * 1e3 / dom_solo["launches"], 2)}}
    else:
        # HBM roofline of the whole forward (the MobileNet models are bandwidth-bound, ~200 FLOP/B): algorithmic bytes =
        # every layer-boundary activation once out and once in (55.2 M elements per frame at 320) + the
        # packed weights once per step; the library's own per-launch accounting (`kernels[*].gbyte`, inputs + weights + outputs
        # of every launch) is given beside it, and the dominant family's own rate.
        act = MOBILENET_ACT_MELEMS * 1e6 * (S / 320.0) ** 2 * es * B
        wbytes = float(eng.weights.numel())
        alg = act + wbytes
        ach = alg / (fwd_ms * 1e-3) / 1e9
        lib_bytes = sum(f["bytes"] for f in prod)
        roofline = {"bound": "hbm", "kernel": "whole forward (%d launches)" % sum(f["launches"] for f in prod), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "traffic_source": "no PMC pass in this run (profiles/r04_cfg4 holds the rocprofv3 passes)",
                    "algorithmic_bytes_per_step": int(alg), "algorithmic": "%.1f M activation elements x %d B x batch %d + %.1f MB of weights" % (MOBILENET_ACT_MELEMS * (S / 320.0) ** 2, es, B, wbytes / 1e6),
                    "per_launch_accounting_gbyte_per_step": round(lib_bytes / 1e9, 3),
                    "mfma_side": {"achieved_tflops": round(gflop_step / fwd_ms, 2), "frac_of_%g" % PEAK_TFLOPS[args.dtype]: round(gflop_step / fwd_ms / PEAK_TFLOPS[args.dtype], 4)},
                    "dominant_family": {"name": dom["name"], "ms": round(dom["ms"], 4), "gbyte_per_s": round(dom["bytes"] / 1e9 / (dom["ms"] * 1e-3), 1),
                                        "tflop_per_s": round(dom["flops"] / 1e12 / (dom["ms"] * 1e-3), 1)}}

    # ---- config 4, the streamed mode ("test_video-style stream"): uint8 frames come from the host, detections go back ----------
    stream_blk = None
    if not trn and args.stream and not args.no_detect:
        NSL = 3
        while NSL % NF:
            NSL += 1
        stream_blk = streamed_block([eng] + [eng.clone() for _ in range(NF - 1)], launch_mode, NF, NSL, B, None, args, fps, world, tdist, torch, dev, pri, False)

    if rank != 0:
        tdist.barrier()
        return
    line = {
        "metric": "frames/sec/GPU @%dx%d %s" % (S, S, "TRN 4-frame clips, ssd4scale_vgg" if trn else "dualrefinedet_mobilenet"),
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE config #%d: %s %dx%d, %s, %s per GPU, forward%s, synthetic VOC-shaped frames + synthetic weights" % (
                       args.config, name, S, S, args.dtype, (("%d clips x 4 frames (per step: 1 static forward over the %d key frames, ONE temporal forward over the %d frames with the key frames' offsets, one Detect call)" % (B, B, 4 * B)) if batched else
                                                       ("%d clips x 4 frames (per step: 1 static + 4 temporal forwards of %d frames + 4 Detect calls, frame by frame as evaluate_trn.py)" % (B, B))) if trn else "batch %d" % B,
                       "" if args.no_detect else " + Detect(top_k 200, conf 0.01, nms 0.45)") + ("" if NCLS == 21 else "; %d classes (not BASELINE's 21; the GFLOP figures stay the 21-class ones)" % NCLS),
                   "global_batch": world * frames_per_step, "parallelism": "%s-sharded x%d, no per-frame collective" % ("clip" if trn else "frame", world),
                   "launch": launch_mode, "launch_calibration_frames_per_s": launch_cal, "resident_batches": NB, "steps_in_flight": NF},
        "repetitions": {"n": len(reps), "reported": "median", "timed_steps_total": len(reps) * args.steps, "ms_per_step_min": round(reps[0] / args.steps * 1e3, 4), "ms_per_step_max": round(reps[-1] / args.steps * 1e3, 4)},
        "fps_per_gpu": round(fps / world, 2), "forward_only_ms_per_step": round(fwd_ms, 4), "forward_tflops": round(gflop_step / fwd_ms, 2),
        "build": build, "roofline": roofline, "kernels": kernels,
        "n_ranks_seen": int(torch.distributed.get_world_size()) if tdist.active() else world, "weights_broadcast_ms": round(bcast_ms, 2) if tdist.active() else None,
        "dist_backend": torch.distributed.get_backend() if tdist.active() else None,
        "numa_pin": tdist.verify_pin(NUMA_PIN, local_rank),
    }
    if one_at_a_time is not None:
        line["one_step_at_a_time"] = one_at_a_time
    if trn:
        line["clips_per_s"] = round(fps / FPC, 2)
        line["trn_static_net_overlapped"] = bool(batched and args.trn_overlap)
        line["trn_mode"] = args.trn_mode      # batched: one temporal forward per step (key-frame offsets broadcast); frames: one per frame index
        if frame_loop is not None:
            line["frame_by_frame"] = frame_loop
    if stream_blk is not None:
        line["stream"] = stream_blk
    # ---- parity of the timed dtype against the fp32 CPU oracle on one frame / one clip (decoded boxes, scores) ----
    if not args.no_parity:
        from oracle import net_ref
        from oracle import oracle as orc
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        pr = pri.cpu().numpy()
        if trn:
            clip = synth.synth_frames(FPC, S, seed=5)
            r_loc, r_conf, r_maps = net_ref.ssd4scale_vgg_forward(sd_s, clip[:1], NCLS, "test", False, False, ret_loc=True)
            t_loc, t_conf, r_offs = net_ref.ssd4scale_vgg_forward(sd_t, clip[:1], NCLS, "test", False, True, ref_loc=r_maps, ret_off=True)
            t1_loc, t1_conf = net_ref.ssd4scale_vgg_forward(sd_t, clip[1:2], NCLS, "test", False, True, offset_list=r_offs)[:2]
            xc = torch.from_numpy(clip).to(dev)
            g_loc, _, g_maps = stat(xc[:1], ret_loc=True)
            gt_loc, gt_conf, g_offs = temp(xc[:1], ref_loc=g_maps, ret_off=True)
            g1_loc, g1_conf = temp(xc[1:2], offset_list=g_offs)[:2]
            def boxes(arm, loc):
                return orc.decode(loc[0], orc.center_size(orc.decode(arm[0], pr)))
            rb0, rb1 = boxes(r_loc.numpy(), t_loc.numpy()), boxes(r_loc.numpy(), t1_loc.numpy())
            gb0, gb1 = boxes(g_loc.cpu().numpy(), gt_loc.cpu().numpy()), boxes(g_loc.cpu().numpy(), g1_loc.cpu().numpy())
            eb = np.abs(np.concatenate([gb0 - rb0, gb1 - rb1]))
            esc = np.abs(np.concatenate([gt_conf.cpu().numpy() - t_conf.numpy().reshape(gt_conf.shape), g1_conf.cpu().numpy() - t1_conf.numpy().reshape(g1_conf.shape)]))
        else:
            x1 = synth.synth_frames(1, S, seed=5)
            r_arm, _, r_odm, r_conf = net_ref.drn_mobilenet_forward(sd, x1, NCLS, True)
            o = net(torch.from_numpy(x1).to(dev))
            rb = orc.decode(r_odm.numpy()[0], orc.center_size(orc.decode(r_arm.numpy()[0], pr)))
            gb = orc.decode(o[2].cpu().numpy()[0], orc.center_size(orc.decode(o[0].cpu().numpy()[0], pr)))
            eb = np.abs(gb - rb)
            esc = np.abs(o[3].cpu().numpy() - r_conf.numpy().reshape(o[3].shape))
        line["parity"] = {"reference": "fp32 CPU oracle (oracle/net_ref.py + oracle/tdrn_oracle.c)", "box_unit": "normalised image coordinates",
                          args.dtype: {"box_linf": float(eb.max()), "box_p999": float(np.quantile(eb, 0.999)), "box_mean": float(eb.mean()),
                                       "score_linf": float(esc.max()), "score_mean": float(esc.mean()),
                                       "note": "all rows (rows next to a sampling discontinuity of the deformable heads included)"}}
        line["box_linf"], line["score_linf"] = float(eb.max()), float(esc.max())
    if world == 1 and not args.no_cpu_baseline:
        from oracle import net_ref
        from oracle import oracle as orc
        cores = min(16, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        pr = pri.cpu().numpy()
        n = max(1, args.cpu_frames // (8 if trn else 2))
        t0 = time.perf_counter()
        done = 0
        if trn:
            sdt_s = {k: torch.from_numpy(v) for k, v in sd_s.items()}
            sdt_t = {k: torch.from_numpy(v) for k, v in sd_t.items()}
            clip = synth.synth_frames(FPC, S, seed=0)
            while done < max(1, n // FPC):
                r_loc, _, r_maps = net_ref.ssd4scale_vgg_forward(sdt_s, clip[:1], NCLS, "test", False, False, ret_loc=True)
                offs = None
                for f in range(FPC):
                    if f == 0:
                        l, c, offs = net_ref.ssd4scale_vgg_forward(sdt_t, clip[:1], NCLS, "test", False, True, ref_loc=r_maps, ret_off=True)
                    else:
                        l, c = net_ref.ssd4scale_vgg_forward(sdt_t, clip[f:f + 1], NCLS, "test", False, True, offset_list=offs)[:2]
                    orc.detect(l.numpy(), c.numpy(), pr, r_loc.numpy(), (500, 375, 500, 375), num_classes=NCLS)
                done += 1
            frames_done = done * FPC
        else:
            sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
            xs = synth.synth_frames(1, S, seed=0)
            while done < n:
                arm, _, odm, conf = net_ref.drn_mobilenet_forward(sdt, xs, NCLS, True)
                orc.detect(odm.numpy(), conf.numpy(), pr, arm.numpy(), (500, 375, 500, 375), num_classes=NCLS)
                done += 1
            frames_done = done
        dtc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(frames_done / dtc, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                                "sample": "%d frames of the same workload, batch 1, fp32 (oracle/: torch-CPU convs + C deformable conv + C Detect), %.1f s" % (frames_done, dtc)}
    print(json.dumps(line))
    tdist.barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)         # >= 100 timed iterations (x --reps repetitions)
    ap.add_argument("--warmup", type=int, default=11)           # evaluate.py:463 drops the first 11 frames
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configuration (2 = the headline)")
    ap.add_argument("--trn-overlap", type=int, default=1, help="config 5, batched mode: 1 = the static net's forward runs on a second stream beside the temporal net's trunk")
    ap.add_argument("--no-frame-loop", action="store_true", help="config 5, batched mode: skip the side measurement of the frame-by-frame order")
    ap.add_argument("--trn-mode", default="batched", choices=["batched", "frames"],
                    help="config 5: 'batched' = one temporal forward over all frames of the step's clips (key-frame offsets broadcast); 'frames' = one temporal forward per frame index, the reference loop's order")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-detect", action="store_true", help="time the network forward only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the box / score error and detection-agreement blocks (oracle forwards on the host)")
    ap.add_argument("--no-modes", action="store_true", help="skip timing the other precisions")
    ap.add_argument("--streams", type=int, default=1, help="split each step's batch over this many concurrent HIP streams")
    ap.add_argument("--launch", choices=["auto", "fixed", "eager"], default="auto",
                    help="two steps in flight: auto = time hipGraph replays AND eager pipelines (streams picked by calibration) briefly and run the timed region with the faster (reported in config.launch); "
                         "eager = the eager pipelines without asking; fixed = what --graph says (replays, or one eager step at a time)")
    ap.add_argument("--graph", type=int, default=1, help="1: the step (forward + Detect) is one captured hipGraph replay; 0: eager launches")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="whole steps in flight (tdrn_amd.engine.InFlight): resident batch j runs on pipeline j %% N (own engine handle, workspace, "
                         "stream and hipGraph; shared weights), so one step's latency-bound tail runs under the next step's trunk; 1 = one step at a time")
    ap.add_argument("--reps", type=int, default=9, help="the K-step loop is timed this many times; the MEDIAN repetition is reported")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the steps cycle through")
    ap.add_argument("--stream", type=int, default=1, help="1: also time the streamed mode (pinned uint8 frames H2D -> preprocess -> net -> Detect -> D2H, two slots in flight) and report it beside the resident figure")
    ap.add_argument("--per-op", action="store_true", help="print per-launch timings of one profiled forward to stderr")
    ap.add_argument("--cpu-frames", type=int, default=96)      # ~16 s of CPU work on the GPU box host
    ap.add_argument("--classes", type=int, default=21, help="num_classes of the nets and of Detect: 21 = VOC (BASELINE's configs), 31 = VID (the TRN drivers), 81 = COCO")
    args = ap.parse_args()
    global NCLS
    NCLS = args.classes
    preset = CONFIGS[args.config]
    for k in ("size", "dtype", "batch"):
        if getattr(args, k) is None:
            setattr(args, k, preset[k])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                                  # (does not return)
    pin_rank_to_numa_node()                                     # before anything touches a GPU
    if args.config in (4, 5):
        return main_other(args)

    import torch
    from tdrn_amd import _lib
    from tdrn_amd import dist as tdist
    from tdrn_amd.data import mb_cfg
    from tdrn_amd.layers import Detect, PriorBox
    from tdrn_amd.model.dualrefinedet_vggbn import build_net
    from tdrn_amd.utils import synth

    rank, local_rank, world = tdist.init()
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    build = _lib.lib().tdrn_version().decode()

    net = build_net("test", args.size, NCLS, 1024, 1, True, True)
    net.set_compute_dtype(args.dtype)
    if os.environ.get("TDRN_BENCH_PLAN_FLAGS"):          # (A/B runs of a kernel choice: tdrn_hip.h TDRN_PLAN_*)
        net.set_plan_flags(int(os.environ["TDRN_BENCH_PLAN_FLAGS"]))
    if rank == 0:
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    t_bc = time.perf_counter()
    if tdist.active():                                           # (N > 1, or TDRN_DIST_FORCE_GROUP=1: the RCCL path at world size 1)
        eng = net.adopt_broadcast_weights(src=0, device=dev)     # the one collective: weights over xGMI
    else:
        eng = net.engine(dev)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t_bc) * 1e3                # (N > 1: rank 0's pack + the broadcast + every rank's adoption)
    B = args.batch
    NB = max(1, args.batches)
    # NB distinct batches, all resident in HBM before the timed region; step k runs batch k % NB
    xb = [torch.from_numpy(synth.synth_frames(B, args.size, seed=100 + rank + 1000 * j)).to(dev) for j in range(NB)]
    x = xb[0]
    pri = PriorBox(mb_cfg["VOC_320" if args.size == 320 else "VOC_512_RefineDet"]).forward().to(dev)
    det = Detect(NCLS, 0, 200, 0.01, 0.45)
    scale = [500.0, 375.0, 500.0, 375.0]

    # optional intra-GPU concurrency: the batch is cut into --streams slices, each with its own engine
    # (shared weight blob), workspace and HIP stream, so that one slice's kernels fill the CUs another
    # slice's tail leaves idle and Detect overlaps the other slices' convolutions.
    NS = max(1, args.streams)
    engines, streams, dets = [eng], [torch.cuda.current_stream(dev)], [det]
    if NS > 1:
        from tdrn_amd.engine import NetEngine
        for i in range(1, NS):
            e2 = NetEngine(**dict(net._engine_args, dtype=args.dtype))
            e2.share_weights(eng)
            engines.append(e2)
            streams.append(torch.cuda.Stream(dev))
            dets.append(Detect(NCLS, 0, 200, 0.01, 0.45))

    def eager_step(k):
        xs = list(torch.chunk(xb[k % NB], NS))
        outs = []
        if NS > 1:
            start = torch.cuda.Event()
            start.record(streams[0])
        for i in range(NS):
            with torch.cuda.stream(streams[i]):
                if NS > 1 and i > 0:
                    streams[i].wait_event(start)
                r = engines[i].forward(xs[i])
                outs.append(r["conf"] if args.no_detect else
                            dets[i].forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale))
        for i in range(1, NS):
            streams[0].wait_stream(streams[i])
        return outs

    def make_stepper(engine, detect):
        """step(k) for one engine: with --graph, one captured hipGraph per resident batch (the batch lives in the graph's
        input buffer: no per-step copy), replayed in turn."""
        def one_step(xin):
            r = engine.forward(xin)
            return r["conf"] if args.no_detect else detect.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
        if args.graph and NS == 1:
            from tdrn_amd.engine import GraphedCall
            graphs = [GraphedCall(one_step, xb[j]) for j in range(NB)]
            return lambda k: graphs[k % NB](graphs[k % NB].inputs[0])
        return lambda k: one_step(xb[k % NB])

    # The default schedule (round 5): TWO steps in flight.  Step k replays resident batch k % NB on pipeline (k % NB) % NF; the timed
    # region is still K steps between two device-wide synchronisations.  `roofline` below comes from single-pipeline profiling
    # passes of engine 0 (a launch's duration with another step's kernels beside it says nothing about the kernel).
    NF = max(1, args.in_flight) if ((args.graph or args.launch == "eager" or os.environ.get("TDRN_BENCH_EAGER_IN_FLIGHT")) and NS == 1) else 1
    while NB % NF:
        NF -= 1

    def in_flight_stepper(engine, as_graph):
        from tdrn_amd.engine import InFlight

        def make_step(e):
            d = Detect(NCLS, 0, 200, 0.01, 0.45)

            def one_step(xin):
                r = e.forward(xin)
                return r["conf"] if args.no_detect else d.forward(r["odm_loc"], r["conf"], pri, arm_loc_data=r["arm_loc"], scale=scale)
            return one_step
        fl = InFlight(make_step, engine, xb, n=NF, graph=as_graph)
        KEEP_ALIVE.append(fl)
        return fl

    def timed(stepper, steps, reps):
        """`reps` repetitions of the K-step loop, each bracketed by barrier + synchronize on both sides and reduced with
        MAX over ranks; returns the per-repetition times (s)."""
        out = []
        for _ in range(reps):
            tdist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                stepper(k)
            torch.cuda.synchronize()
            tdist.barrier()
            out.append(tdist.max_over_ranks(time.perf_counter() - t0, dev))
        return out

    launch_mode, launch_cal = ("hipGraph replay" if (args.graph and NS == 1) else "eager"), None
    if NF > 1:
        flight = in_flight_stepper(eng, bool(args.graph) and args.launch != "eager")
        if args.launch == "eager":
            for k in range(4):
                flight.launch(k)
            flight.pick_streams()
            launch_mode = "eager"
        elif args.graph and args.launch == "auto":
            flight, launch_mode, launch_cal = _choose_launch(flight, lambda: in_flight_stepper(eng, False), args.steps, world * B, tdist, torch, dev)
        elif not args.graph:
            launch_mode = "eager"
        step = flight.launch
        engines = flight.engines
    else:
        step = make_stepper(eng, det) if NS == 1 else eager_step
    for k in range(args.warmup):
        step(k)
    reps = sorted(timed(step, args.steps, max(1, args.reps)))
    for e_ in engines:
        e_.check()                  # tdrn_net_check: a device-side hand-off that timed out inside the replays above fails the run HERE
    dt = reps[len(reps) // 2]                                    # the median repetition
    fps = world * B * args.steps / dt
    one_at_a_time = None
    if NF > 1:
        # the same steps one at a time (one pipeline, one hipGraph per resident batch), timed in the same process
        st1 = make_stepper(eng, det)
        for k in range(3):
            st1(k)
        r1 = sorted(timed(st1, args.steps, 3))
        one_at_a_time = {"frames_per_s": round(world * B * args.steps / r1[1], 2), "ms_per_step": round(r1[1] / args.steps * 1e3, 4), "repetitions": 3}
        KEEP_ALIVE.append(st1)

    # ---- forward-only split and per-kernel roofline (separate, event-instrumented passes) ---------
    def forward_ms(engine):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            engine.forward(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / 5 * 1e3
    fwd_ms = forward_ms(eng)

    def profiled(engine, mode):
        engine.set_profile(mode)
        engine.forward(x)
        engine.forward(x)
        torch.cuda.synchronize()
        st = engine.kernel_stats()
        ops = engine.op_stats() if args.per_op else None
        engine.set_profile(0)
        return st, ops

    def conv_family(stats):
        """the 3x3 MFMA conv family of a step (conv3x3_patch.hip + conv3x3_pp.hip launches are accounted together)"""
        return max(stats, key=lambda s: s["ms"])
    stats, ops = profiled(eng, 1)                            # every launch alone on one stream
    stats_prod, ops_prod = profiled(eng, 2)                  # the production schedule (side lanes on)
    if ops and rank == 0:
        prod = {o["name"]: o["ms"] for o in ops_prod}
        print("%-44s %11s %11s %8s %9s %8s" % ("launch", "alone us", "in step us", "GFLOP", "TFLOP/s", "GB"), file=sys.stderr)
        for o in ops:
            tf = o["flops"] / (o["ms"] * 1e-3) / 1e12 if o["ms"] > 0 else 0.0
            print("%-44s %11.1f %11.1f %8.1f %9.1f %8.3f" % (o["name"], o["ms"] * 1e3, prod.get(o["name"], 0.0) * 1e3, o["flops"] / 1e9, tf, o["bytes"] / 1e9), file=sys.stderr)
    conv = conv_family(stats_prod)                           # the dominant kernel family of the step
    solo = next(s for s in stats if s["name"] == conv["name"])
    peak = PEAK_TFLOPS[args.dtype]
    achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
    achieved_solo = solo["flops"] / (solo["ms"] * 1e-3) / 1e12 if solo["ms"] > 0 else 0.0
    traffic, traffic_src = pmc_traffic(conv["name"], args, build)
    # `achieved` / `frac` are the PRODUCTION figures: hipEvents on the stream each launch runs on, side lanes on -- what
    # rocprofv3 --kernel-trace sees in the timed loop.  The single-stream figures are given beside them.
    roofline = {"bound": "mfma", "kernel": conv["name"], "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": conv["launches"], "gflop_per_launch": round(conv["flops"] / 1e9 / conv["launches"], 2),
                "us_per_launch": round(conv["ms"] * 1e3 / conv["launches"], 2), "mode": "production schedule (side lanes on)",
                "single_stream": {"achieved": round(achieved_solo, 2), "frac": round(achieved_solo / peak, 4),
                                  "us_per_launch": round(solo["ms"] * 1e3 / solo["launches"], 2)}}
    solo_ms = {s["name"]: s["ms"] for s in stats}
    kernels = {s["name"]: {"ms": round(s["ms"], 4), "ms_single_stream": round(solo_ms.get(s["name"], 0.0), 4), "launches": s["launches"],
                           "gflop": round(s["flops"] / 1e9, 3), "gbyte": round(s["bytes"] / 1e9, 4)} for s in stats_prod}

    # ---- the streamed mode: frames come from the host, detections go back (test_video.py:98-115 as a pipeline) -----------
    # (before the other precisions' engines are created and destroyed: on ROCm 7.2 a hipGraph captured for the fp32 engine AFTER
    # other engines' graphs and events had been destroyed crashed in hipGraphLaunch)
    stream_blk = None
    if NS == 1 and args.stream and not args.no_detect:          # (N > 1: every rank feeds its own GPU from its own pinned buffers)
        NSL = max(3, NB)
        while NSL % NF:
            NSL += 1
        stream_blk = streamed_block([eng] + [eng.clone() for _ in range(NF - 1)], launch_mode, NF, NSL, B, None, args, fps, world, tdist, torch, dev, pri, True)

    # ---- the other precisions of the same workload, timed in this run (N = 1 only) -----------------------------------
    modes = None
    if world == 1 and NS == 1 and not args.no_modes:
        modes = {}
        for dtm in [d for d in ("fp16", "fp32", "bf16") if d != args.dtype]:
            print("bench.py: timing the %s mode" % dtm, file=sys.stderr, flush=True)
            net.set_compute_dtype(dtm)
            e2 = net.engine(dev)
            if NF > 1:
                fl2 = in_flight_stepper(e2, launch_mode != "eager")      # (the launch mode the headline chose)
                if launch_mode == "eager":
                    for k in range(4):
                        fl2.launch(k)
                    fl2.pick_streams()
                st2 = fl2.launch
            else:
                st2 = make_stepper(e2, Detect(NCLS, 0, 200, 0.01, 0.45))
            k_steps = max(5, min(args.steps, 10 if dtm == "fp32" else args.steps))
            for k in range(3):
                st2(k)
            r2 = sorted(timed(st2, k_steps, 3))
            t2 = r2[len(r2) // 2]
            # The conv family of a side mode is timed with every launch ALONE on one stream (profile mode 1): the production-schedule
            # figure of an engine depends on which hardware queues its side lanes were given when it was built (the in-flight stepper
            # above takes lanes from the pool first), which made a side mode's `frac` incomparable with the headline's (round-5 review).
            # Compare with `roofline.single_stream` of the headline.
            sp, _ = profiled(e2, 1)
            c2 = next(s_ for s_ in sp if s_["name"] == conv["name"])
            a2 = c2["flops"] / (c2["ms"] * 1e-3) / 1e12 if c2["ms"] > 0 else 0.0
            modes[dtm] = {"frames_per_s": round(B * k_steps / t2, 2), "ms_per_step": round(t2 / k_steps * 1e3, 4), "steps": k_steps,
                          "repetitions": 3, "forward_only_ms_per_step": round(forward_ms(e2), 4),
                          "roofline_single_stream": {"kernel": c2["name"], "achieved": round(a2, 2), "peak": PEAK_TFLOPS[dtm], "unit": "TFLOP/s",
                                                     "frac": round(a2 / PEAK_TFLOPS[dtm], 4),
                                                     "mode": "every launch alone on one stream (= roofline.single_stream of the headline dtype)"}}
            KEEP_ALIVE.extend((st2, e2))
        net.set_compute_dtype(args.dtype)

    if rank == 0:
        line = {
            "metric": "frames/sec/GPU @%dx%d dualrefinedet_vggbn; box Linf vs CPU ref" % (args.size, args.size),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "dualrefinedet_vggbn %dx%d multihead, %s, batch %d per GPU, forward%s, synthetic "
                                   "VOC-shaped frames + synthetic weights%s" %
                                   (args.size, args.size, args.dtype, B, "" if args.no_detect else " + Detect(top_k 200, conf 0.01, nms 0.45)",
                                    "" if NCLS == 21 else ", %d classes (not BASELINE's 21)" % NCLS),
                       "global_batch": world * B, "parallelism": "frame-sharded x%d, no per-frame collective" % world,
                       "launch": launch_mode,
                       "launch_calibration_frames_per_s": launch_cal,
                       "eager_stream_calibration": (getattr(flight, "stream_calibration", None) or {}).get("picked") if launch_mode == "eager" else None,
                       "resident_batches": NB,
                       "steps_in_flight": NF,
                       "schedule": ("%d steps in flight: resident batch j runs on pipeline j %% %d (own engine handle, workspace, HIP stream%s; one weight blob); "
                                    "K steps between two device-wide synchronisations" % (NF, NF, ", hipGraph" if launch_mode != "eager" else ", launched eagerly")) if NF > 1 else "one step at a time"},
            "repetitions": {"n": len(reps), "reported": "median", "timed_steps_total": len(reps) * args.steps, "ms_per_step_min": round(reps[0] / args.steps * 1e3, 4),
                            "ms_per_step_max": round(reps[-1] / args.steps * 1e3, 4)},
            "fps_per_gpu": round(fps / world, 2),
            "forward_only_ms_per_step": round(fwd_ms, 4),
            "forward_tflops": round((GFLOP_PER_FRAME.get(args.size, 0) + head_gflop_delta(args.size)) * B / fwd_ms, 2),
            "build": build,
            "roofline": roofline,
            "kernels": kernels,
            # N > 1: what the first hardware run should tell -- how many ranks really joined, what the one collective cost, where
            # each rank's feeder threads were pinned (tdrn_amd/dist.py pin_to_gpu_numa_node; rank 0's record)
            "n_ranks_seen": int(torch.distributed.get_world_size()) if tdist.active() else world,
            "weights_broadcast_ms": round(bcast_ms, 2) if tdist.active() else None,
            "dist_backend": torch.distributed.get_backend() if tdist.active() else None,
            "numa_pin": tdist.verify_pin(NUMA_PIN, local_rank),
        }
        if one_at_a_time is not None:
            line["one_step_at_a_time"] = one_at_a_time
            line["roofline"]["mode"] += "; single-pipeline profiling passes of engine 0 (`value` is timed with %d steps in flight)" % NF
        if modes is not None:
            line["modes"] = modes
        if stream_blk is not None:
            line["stream"] = stream_blk
        if not args.no_parity:
            others = [d for d in ("fp32", "fp16", "bf16") if d != args.dtype]
            par = parity_vs_oracle(args.size, [args.dtype] + others, dev)
            line["parity"] = par
            line["box_linf"] = par[args.dtype]["box_linf"]              # the timed dtype's figure; fp32 mode: par["fp32"]
            line["score_linf"] = par[args.dtype]["score_linf"]
            if world == 1:
                line["detections"] = detection_agreement(args.size, [args.dtype] + others, dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.size, args.cpu_frames)
        print(json.dumps(line))
    tdist.barrier()


if __name__ == "__main__":
    main()
