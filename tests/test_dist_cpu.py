"""world_size-2 gloo tests (CPU) of the N>1 path: frame sharding covers every frame exactly once,
the weight blob arrives by one broadcast, per-rank results gather back in frame order, and the
max-over-ranks timing reduction works.  The data path itself has no collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tdrn_amd import dist as tdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = tdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    # weight blob: rank 0 "packs", the others receive
    blob = torch.arange(1000, dtype=torch.uint8) if rank == 0 else torch.zeros(1000, dtype=torch.uint8)
    tdist.broadcast_blob(blob, src=0)
    assert torch.equal(blob, torch.arange(1000, dtype=torch.uint8))
    frames = np.arange(n_frames, dtype=np.float32) * 10
    sl = tdist.shard_slice(n_frames, rank, world)
    local = frames[sl] + 1.0                      # stand-in for per-frame independent inference
    got = tdist.gather_results((sl.start, local), rank, world, dst=0)
    # the data-plane form: one tensor gather of the (b_local, C, top_k, 5)-shaped detections, ragged over the ranks
    det = torch.from_numpy(local).reshape(-1, 1, 1, 1).repeat(1, 2, 3, 5)
    gt = tdist.gather_results(det, rank, world, dst=0)
    if rank == 0:
        assert [int(g.shape[0]) for g in gt] == [len(range(n_frames)[tdist.shard_slice(n_frames, k, world)]) for k in range(world)]
        assert torch.equal(torch.cat(gt, 0)[:, 0, 0, 0], torch.from_numpy(frames + 1.0))
    else:
        assert gt is None
    t = tdist.max_over_ranks(1.0 + rank)
    assert t == float(world)
    tdist.barrier()
    if rank == 0:
        out = np.concatenate([g[1] for g in sorted(got, key=lambda g: g[0])])
        q.put(out)
    dist.destroy_process_group()


def _run_sharding(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(out, np.arange(n_frames, dtype=np.float32) * 10 + 1.0)


@pytest.mark.parametrize("n_frames", [7, 8, 1])
def test_two_rank_frame_sharding(n_frames):
    _run_sharding(2, n_frames)


@pytest.mark.parametrize("n_frames", [256, 13, 5])
def test_eight_rank_frame_sharding(n_frames):
    """The driver's first multi-GPU run is N = 8: the same worker at world size 8 (gloo, CPU) -- 256 frames = bench.py's global
    batch at 8 x 32, 13 = ragged shards, 5 = three ranks hold NOTHING (empty slices gather as empty tensors)."""
    _run_sharding(8, n_frames)


def test_shard_helpers_partition_exactly():
    for n in (0, 1, 5, 32, 33):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                s = tdist.shard_slice(n, r, world)
                seen += list(range(n))[s]
            assert seen == list(range(n))
            rr = sorted(i for r in range(world) for i in tdist.shard_stream(n, r, world))
            assert rr == list(range(n))
            sizes = [len(range(n)[tdist.shard_slice(n, r, world)]) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker_mismatch(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    tdist.init(backend="gloo")
    res = []
    # (1) one rank passes a tensor, the other None (an "empty shard"): both must RAISE, not hang in different collectives
    try:
        tdist.gather_results(torch.zeros(2, 3) if rank == 0 else None, rank, world)
        res.append("no error")
    except ValueError as e:
        res.append("raised" if "disagree" in str(e) else str(e))
    # (2) trailing shapes differ: raises on every rank
    try:
        tdist.gather_results(torch.zeros(2, 3 + rank), rank, world)
        res.append("no error")
    except ValueError as e:
        res.append("raised" if "trailing shape" in str(e) else str(e))
    # (3) the right way to gather an empty shard: an empty tensor of the right trailing shape; a 0-dim tensor travels as an object
    got = tdist.gather_results(torch.ones(2, 3) if rank == 0 else torch.zeros(0, 3), rank, world)
    if rank == 0:
        res.append([tuple(g.shape) for g in got])
    sc = tdist.gather_results(torch.tensor(float(rank)), rank, world)
    if rank == 0:
        res.append([float(v) for v in sc])
    # (4) ADVICE r04: a tensor the function cannot carry on ONE rank (a dtype outside its table) raises on EVERY rank, after the
    # header exchange -- not on that rank alone before it, which left the others inside the all_gather
    try:
        tdist.gather_results(torch.zeros(2, 3, dtype=torch.complex64) if rank == 1 else torch.zeros(2, 3), rank, world)
        res.append("no error")
    except ValueError as e:
        res.append("raised" if "unsupported tensor on ranks [1]" in str(e) else str(e))
    tdist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def test_gather_results_mismatch_raises_on_every_rank():
    """the collective sequence must not depend on each rank's local Python type."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mismatch, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][:2] == ["raised", "raised"] and got[1][:2] == ["raised", "raised"]
    assert got[0][2] == [(2, 3), (0, 3)] and got[0][3] == [0.0, 1.0]
    assert got[0][4] == "raised" and got[1][2] == "raised"


def _fake_sysfs(root, gpus, nodes):
    """gpus: [(bdf, vendor, class, numa_node)], nodes: {node: cpulist}"""
    for bdf, vendor, cls, node in gpus:
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "class").write_text(cls + "\n")
        (d / "numa_node").write_text("%d\n" % node)
    for node, cl in nodes.items():
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")


def test_rank_pinning_reads_the_gpus_numa_node_from_sysfs(tmp_path, monkeypatch):
    """bench.py pins every rank to the CPUs of its GPU's NUMA node before the first GPU call (sysfs only)."""
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    lo, hi = allowed[:half], allowed[half:] or allowed[:half]
    fmt = lambda cpus: ",".join(str(c) for c in cpus)
    _fake_sysfs(tmp_path, [("0000:05:00.0", "0x1002", "0x120000", 0), ("0000:15:00.0", "0x8086", "0x030000", 0),
                           ("0000:85:00.0", "0x1002", "0x120000", 1), ("0000:95:00.0", "0x1002", "0x038000", -1)], {0: fmt(lo), 1: fmt(hi)})
    assert tdist.gpu_numa_nodes(str(tmp_path)) == [("0000:05:00.0", 0), ("0000:85:00.0", 1), ("0000:95:00.0", -1)]
    d0 = tdist.pin_to_gpu_numa_node(0, 4, str(tmp_path), apply=False)
    d1 = tdist.pin_to_gpu_numa_node(1, 4, str(tmp_path), apply=False)
    assert d0["cpus"] == len(lo) and d0["first_cpu"] == lo[0] and "numa node 0" in d0["how"]
    assert d1["cpus"] == len(hi) and d1["first_cpu"] == hi[0] and "numa node 1" in d1["how"]
    d2 = tdist.pin_to_gpu_numa_node(2, 4, str(tmp_path), apply=False)          # no node reported: an even split
    assert "even split" in d2["how"] and d2["cpus"] >= 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")                           # an index remap is followed
    assert "numa node 1" in tdist.pin_to_gpu_numa_node(0, 2, str(tmp_path), apply=False)["how"]
    # applying it really narrows the mask (and is undone for the rest of the test session)
    before = os.sched_getaffinity(0)
    try:
        monkeypatch.delenv("HIP_VISIBLE_DEVICES")
        d = tdist.pin_to_gpu_numa_node(0, 4, str(tmp_path), apply=True)
        assert d["pinned"] and os.sched_getaffinity(0) == set(lo)
    finally:
        os.sched_setaffinity(0, before)


def _forced_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TDRN_DIST_FORCE_GROUP="1")
    r, lr, w = tdist.init(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and tdist.active()
    blob = torch.arange(100, dtype=torch.uint8)
    assert torch.equal(tdist.broadcast_blob(blob.clone()), blob)
    det = torch.randn(3, 2, 4, 5)
    got = tdist.gather_results(det, 0, 1)
    assert len(got) == 1 and torch.equal(got[0], det)
    assert tdist.gather_results("x", 0, 1) == ["x"]
    with pytest.raises(ValueError, match="unsupported tensor"):
        tdist.gather_results(torch.zeros(2, dtype=torch.complex64), 0, 1)
    assert tdist.max_over_ranks(2.5) == 2.5
    tdist.barrier()
    q.put("ok")
    dist.destroy_process_group()


def test_forced_group_at_world_size_one_runs_the_collectives():
    """TDRN_DIST_FORCE_GROUP=1 (how tests/test_gpu_dist.py::test_rccl_* executes the RCCL branches on one GPU): with it a one-rank
    group is built and every collective of tdrn_amd.dist runs; without it world size 1 stays collective-free."""
    assert not tdist.active()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0 and q.get(timeout=5) == "ok"
