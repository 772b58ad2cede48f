"""Frame-sharded multi-GPU inference: one process per GPU, weights replicated by ONE RCCL broadcast
of the packed blob over xGMI at start-up, then every rank runs its own frames with no collective in
the per-frame path.  The reference has no inference-time parallelism at all
(single GPU, batch 1, evaluate.py:452); its only multi-GPU mechanism is nn.DataParallel for
training (train.py:157-158)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1).
    backend: "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo"."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # TDRN_DIST_BACKEND=gloo: several ranks on ONE GPU (tests on a 1-GPU box; RCCL refuses duplicate devices)
            backend = os.environ.get("TDRN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def shard_slice(n, rank, world):
    """Contiguous, balanced slice of n frames for `rank` (clips / videos are never split)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def shard_stream(n, rank, world):
    """Round-robin frame indices r, r+R, ... for a live stream (test_video.py style)."""
    return list(range(rank, n, world))


def broadcast_blob(blob, src=0):
    """The one collective of the path: replicate the packed weight blob."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_results(local, rank, world, dst=0):
    """Host-side gather of per-rank results (e.g. (b_local, C, top_k, 5) detections, 84 kB/frame).
    Returns the list of every rank's object on `dst`, None elsewhere."""
    if world == 1 or not dist.is_initialized():
        return [local]
    out = [None] * world if rank == dst else None
    dist.gather_object(local, out, dst=dst)
    return out


def max_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
