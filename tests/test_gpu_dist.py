"""The multi-GPU path exercised on ONE GPU: two fresh processes (spawned before anything touches the
GPU) share cuda:0 over the gloo backend -- rank 0 packs the weights, rank 1 receives the blob through
EngineModule.adopt_broadcast_weights / NetEngine.broadcast_weights and never sees the checkpoint -- and `bench.py
--gpus 2` runs its N > 1 branch (tdist.init, barrier, max-over-ranks timing, whole-job value) the same way.  On a
real node the backend is RCCL and every rank has its own GPU; the code path is the same."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(argv, world=2, timeout=600):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TDRN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o.decode(), e.decode()))
    for rc, o, e in outs:
        assert rc == 0, "rank failed (rc %d):\n%s\n%s" % (rc, o[-2000:], e[-4000:])
    return outs


def test_broadcast_weights_two_ranks_one_gpu(tmp_path):
    _launch([os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)])
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert r0["wsum"][0] == r1["wsum"][0] and r0["wsum"][0] != 0.0          # the same packed blob on both ranks
    for k in ("arm", "odm", "conf"):
        assert np.array_equal(r0[k], r1[k]), k                                # rank 1 never loaded a state_dict
    # frame sharding: contiguous halves, each equal to the corresponding rows of the full batch
    assert r0["shard"].tolist() == [0, 2] and r1["shard"].tolist() == [2, 4]
    assert np.array_equal(r0["shard_odm"], r0["odm"][0:2]) and np.array_equal(r1["shard_odm"], r0["odm"][2:4])


def test_bench_py_two_ranks_one_gpu():
    outs = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "4",
                    "--no-cpu-baseline"])
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][1].splitlines() if l.startswith("{")]    # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 8 and d["value"] > 0
    assert abs(d["value"] - 2 * 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3            # whole-job frames / max-over-ranks time
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0


def test_bench_py_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun variables in the environment -- the driver's command form -- starts two
    fresh ranks itself (before the parent touches a GPU), relays rank 0's line and its exit code.  On this 1-GPU box the ranks
    share device 0 over gloo (TDRN_DIST_ONE_DEVICE / TDRN_DIST_BACKEND: test-only switches; on a node every rank has its own
    GPU and the backend is RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TDRN_DIST_BACKEND="gloo", TDRN_DIST_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "3", "--warmup", "2",
                        "--reps", "2", "--no-cpu-baseline", "--no-parity"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["steps"] == 3 and d["value"] > 0
    # a rank that fails makes the launcher fail
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "1", "--size", "333"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode != 0


@pytest.mark.parametrize("config,batch", [(4, 4), (5, 2)])
def test_bench_py_other_configs_launch_their_own_ranks(config, batch):
    """`python bench.py --config N --gpus 2` (the driver's command form, no torchrun variables): two ranks over gloo on this one
    device, weights broadcast from rank 0, the frames / clips sharded by rank, one line from rank 0 with the whole-job rate --
    config 4 with its streamed mode (every rank feeds its own slots), config 5 with the batched temporal forward."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TDRN_DIST_BACKEND="gloo", TDRN_DIST_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--gpus", "2", "--batch", str(batch), "--steps", "2",
                        "--warmup", "1", "--reps", "2", "--no-cpu-baseline", "--no-parity"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    per_rank = batch * (4 if config == 5 else 1)
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["config"]["global_batch"] == 2 * per_rank and d["value"] > 0
    assert abs(d["value"] - 2 * per_rank * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    if config == 4:
        assert d["stream"]["detections_identical_to_unstreamed"] and d["stream"]["frames_per_s"] > 0
    else:
        assert d["trn_mode"] == "batched" and d["frame_by_frame"]["frames_per_s"] > 0


@pytest.mark.parametrize("config,batch", [(3, 2), (4, 4), (5, 2)])
def test_bench_py_other_baseline_configs(config, batch):
    """`bench.py --config N` (BASELINE.json configs 3-5; the driver's default command is config 2) prints the same line shape:
    metric / value / roofline / cpu_baseline / parity, with the roofline bound the workload has (MFMA for the VGG nets and the TRN
    clips, HBM for the MobileNet model) -- run here at a small batch so that it takes seconds."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--batch", str(batch), "--steps", "2",
                        "--warmup", "1", "--reps", "2", "--cpu-frames", "2", "--no-modes", "--stream", "0"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["unit"] == "frames/s" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == ("hbm" if config == 4 else "mfma") and 0 < r["frac"] < 1 and r["peak"] == (8000.0 if config == 4 else 2500.0)
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["box_linf"] < 1.0 and "workload" in d["config"]
    if config == 5:
        assert abs(d["clips_per_s"] * 4 - d["value"]) / d["value"] < 1e-3
        assert d["config"]["global_batch"] == batch * 4 and d["trn_mode"] == "batched" and d["trn_static_net_overlapped"] is True


def _one_rank_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TDRN_DIST_BACKEND")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               TDRN_DIST_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def test_rccl_branches_at_world_size_one(tmp_path):
    """The `nccl` (= RCCL) branches of tdrn_amd/dist.py had run on no GPU before round 6 (every multi-rank test is gloo: RCCL refuses
    two ranks on one device).  One fresh rank with backend "nccl" and TDRN_DIST_FORCE_GROUP=1 drives init, adopt_broadcast_weights /
    broadcast_weights, broadcast_blob, gather_results (device tensor, host tensor, empty shard, object, 0-dim, the two unsupported-tensor
    headers), max_over_ranks and barrier through RCCL on cuda:0 (tests/_rccl_worker.py).  The reference's only multi-GPU code is
    train.py:157-158 (nn.DataParallel)."""
    out = tmp_path / "rccl.npz"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), str(out)], env=_one_rank_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, "rc %d\n%s\n%s" % (p.returncode, p.stdout.decode()[-2000:], p.stderr.decode()[-4000:])
    r = np.load(out)
    assert r["ok"][0] == 1 and r["wsum"][0] != 0.0


def test_bench_py_one_rank_over_rccl():
    """`bench.py --gpus 1` with a forced process group: the bench's N > 1 branch (weights by RCCL broadcast, barriers and the
    max-over-ranks all_reduce around every timed repetition) on the backend the driver's 8-GPU run will use."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "4", "--steps", "3", "--warmup", "2", "--reps", "2",
                        "--no-cpu-baseline", "--no-parity", "--no-modes", "--stream", "0"], env=_one_rank_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["dist_backend"] == "nccl" and d["weights_broadcast_ms"] > 0 and d["value"] > 0
