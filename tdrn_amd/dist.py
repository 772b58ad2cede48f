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
    """Gather of per-rank results on `dst`: returns the list of every rank's result there, None elsewhere.
    A TENSOR (e.g. (b_local, C, top_k, 5) detections, 84 kB/frame) travels as ONE collective on the data plane -- ranks may
    hold different numbers of frames: the leading sizes are exchanged first and the payload padded to the largest -- on the
    tensor's own device with RCCL, through the host with gloo.  Anything else falls back to gather_object (control plane,
    pickled: fine for a few scalars, not for detections -- at 8 x 9k frames/s that would be 6 GB/s of pickling on rank 0)."""
    if world == 1 or not dist.is_initialized():
        return [local]
    if not torch.is_tensor(local):
        out = [None] * world if rank == dst else None
        dist.gather_object(local, out, dst=dst)
        return out
    nccl = dist.get_backend() == "nccl"
    work = local if (nccl or not local.is_cuda) else local.cpu()
    n = torch.tensor([work.shape[0]], dtype=torch.int64, device=work.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(t.item()) for t in sizes]
    nmax = max(sizes)
    if work.shape[0] < nmax:
        pad = torch.zeros((nmax - work.shape[0],) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
        work = torch.cat([work, pad], 0)
    work = work.contiguous()
    bufs = [torch.empty_like(work) for _ in range(world)] if rank == dst else None
    dist.gather(work, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:k].to(local.device) for b, k in zip(bufs, sizes)]


def max_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
