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


@pytest.mark.parametrize("n_frames", [7, 8, 1])
def test_two_rank_frame_sharding(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(out, np.arange(n_frames, dtype=np.float32) * 10 + 1.0)


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
