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


def group_forced():
    """TDRN_DIST_FORCE_GROUP=1: build the process group -- and run every collective of this module -- at world size 1 too.
    That is how the RCCL branches below are executed on a 1-GPU box (tests/test_gpu_dist.py::test_rccl_*, `bench.py --gpus 1`
    with the variable set): a one-rank communicator goes through the same ncclCommInitRank / ncclBroadcast / ncclAllGather /
    ncclAllReduce entry points as an 8-rank one, only without a peer on the other end of xGMI."""
    return os.environ.get("TDRN_DIST_FORCE_GROUP") == "1"


def active():
    """Whether this module's collectives run: a process group exists and has peers (or TDRN_DIST_FORCE_GROUP says so)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or group_forced())


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1 unless TDRN_DIST_FORCE_GROUP=1).
    backend: "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".  Call before the first GPU call of the process."""
    rank, local_rank, world = env_world()
    if (world > 1 or group_forced()) and not dist.is_initialized():
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


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD GPU of this host in PCI-address order (= the HIP enumeration order on a default ROCm setup),
    read from sysfs only -- nothing here touches the GPU or the HIP runtime.  -1 where the platform reports none."""
    base = os.path.join(sysfs, "bus", "pci", "devices")
    out = []
    try:
        devs = sorted(os.listdir(base))
    except OSError:
        return out
    for bdf in devs:
        d = os.path.join(base, bdf)
        try:
            vendor = open(os.path.join(d, "vendor")).read().strip()
            cls = open(os.path.join(d, "class")).read().strip()
        except OSError:
            continue
        # AMD, display controller (0x03xxxx) or processing accelerator (0x12xxxx: Instinct parts)
        if vendor != "0x1002" or not (cls.startswith("0x03") or cls.startswith("0x12")):
            continue
        try:
            node = int(open(os.path.join(d, "numa_node")).read().strip())
        except (OSError, ValueError):
            node = -1
        out.append((bdf, node))
    return out


def pin_to_gpu_numa_node(local_rank, world, sysfs="/sys", apply=True):
    """CPU affinity of this rank = the CPUs of its GPU's NUMA node (at 8 x ~10k frames/s the host feeds
    8 x 18 MB per step over PCIe from 8 Python feeders; a feeder on the far socket pays the inter-socket hop on every
    pinned-buffer write and H2D descriptor).  Call BEFORE the first GPU call of the process (the runtime's helper threads
    inherit the mask).  The GPU is chosen as the local_rank-th AMD device in PCI order, through HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES when they are plain index lists.  Falls back to an even split of the CPUs this process may use
    when sysfs names no node.  Returns a description for the bench line."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"pinned": False, "reason": "no sched_getaffinity on this platform"}
    gpus = gpu_numa_nodes(sysfs)
    idx = local_rank
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                idx = [int(t) for t in v.split(",")][local_rank]
            except (ValueError, IndexError):
                pass
            break
    node, cpus, how = -1, None, None
    if 0 <= idx < len(gpus):
        node = gpus[idx][1]
    if node >= 0:
        try:
            cpus = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")).read()) & allowed
            how = "numa node %d of GPU %s" % (node, gpus[idx][0])
        except OSError:
            cpus = None
    if not cpus:
        order = sorted(allowed)
        per = max(1, len(order) // max(1, world))
        cpus = set(order[(local_rank % max(1, world)) * per:(local_rank % max(1, world) + 1) * per]) or allowed
        how = "even split of %d usable CPUs over %d ranks (sysfs names no NUMA node for the GPU)" % (len(order), world)
    if apply:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError as e:
            return {"pinned": False, "reason": str(e)}
    return {"pinned": bool(apply), "cpus": len(cpus), "first_cpu": min(cpus), "how": how,
            "gpu_bdf_assumed": gpus[idx][0] if 0 <= idx < len(gpus) else None}


def hip_device_bdf(device_index):
    """PCI address of the HIP device (call AFTER the GPU is initialised): what pin_to_gpu_numa_node's sysfs-order guess is checked
    against -- HIP's enumeration order may differ from the PCI address order, and then `numa_pin.how` would claim the wrong node."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:                                                    # (older torch: no PCI fields)
        return None


def verify_pin(pin, device_index):
    """Adds the HIP-side PCI address to a pin_to_gpu_numa_node record and says whether the guess was right."""
    if not isinstance(pin, dict):
        return pin
    pin = dict(pin)
    pin["gpu_bdf_hip"] = hip_device_bdf(device_index)
    a, b = pin.get("gpu_bdf_assumed"), pin["gpu_bdf_hip"]
    pin["bdf_match"] = (a.lower() == b.lower()) if (a and b) else None
    return pin


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
    if active():
        dist.broadcast(blob, src=src)
    return blob


def gather_results(local, rank, world, dst=0):
    """Gather of per-rank results on `dst`: returns the list of every rank's result there, None elsewhere.
    A TENSOR (e.g. (b_local, C, top_k, 5) detections, 84 kB/frame) travels as ONE collective on the data plane -- ranks may
    hold different numbers of frames: the leading sizes are exchanged first and the payload padded to the largest -- on the
    tensor's own device with RCCL, through the host with gloo.  Anything else falls back to gather_object (control plane,
    pickled: fine for a few scalars, not for detections -- at 8 x 9k frames/s that would be 6 GB/s of pickling on rank 0)."""
    if not active():
        return [local]
    nccl = dist.get_backend() == "nccl"
    ctl_dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    # Header first: every rank says WHAT it holds (tensor or object, rank, trailing shape, dtype, leading size), so all ranks
    # take the same collective sequence and a mismatch raises on EVERY rank instead of hanging one side in a different collective.
    is_t = torch.is_tensor(local) and local.dim() >= 1            # (a 0-dim tensor travels as an object)
    DT = [torch.float32, torch.float16, torch.bfloat16, torch.float64, torch.int32, torch.int64, torch.uint8, torch.int8, torch.bool, torch.int16]
    # (built on the host in one go and moved once: twelve element writes into a device tensor are twelve tiny H2D copies under RCCL;
    # kind 2 = "a tensor this function cannot carry": it travels in the header too, so that the rank concerned raises AFTER the
    # all_gather, together with everybody else, instead of leaving the others inside the collective)
    h_host = [0] * 12
    if is_t:
        if local.dim() > 8 or local.dtype not in DT:
            h_host[0], h_host[1] = 2, local.dim()
        else:
            h_host[:4] = [1, local.dim(), DT.index(local.dtype), local.shape[0]]
            h_host[4:4 + local.dim() - 1] = [int(d) for d in local.shape[1:]]
    hdr = torch.tensor(h_host, dtype=torch.int64).to(ctl_dev)
    hdrs = [torch.zeros_like(hdr) for _ in range(world)]
    dist.all_gather(hdrs, hdr)
    hdrs = [h.cpu().tolist() for h in hdrs]
    if any(h[0] == 2 for h in hdrs):
        raise ValueError("gather_results: unsupported tensor on ranks %r (more than 8 dimensions or a dtype outside %r)"
                         % ([r for r, h in enumerate(hdrs) if h[0] == 2], [str(d) for d in DT]))
    kinds = {h[0] for h in hdrs}
    if len(kinds) != 1:
        raise ValueError("gather_results: ranks disagree on what they gather (tensor on ranks %r, object on the others); pass an "
                         "empty tensor of the right trailing shape for an empty shard" % [r for r, h in enumerate(hdrs) if h[0]])
    if not is_t:
        out = [None] * world if rank == dst else None
        dist.gather_object(local, out, dst=dst)
        return out
    if any(h[1:3] != hdrs[0][1:3] or h[4:] != hdrs[0][4:] for h in hdrs):
        raise ValueError("gather_results: trailing shape / dtype differ across ranks: %r" % [(h[1], h[2], h[4:4 + max(0, h[1] - 1)]) for h in hdrs])
    # payload: on the data plane -- RCCL wants device tensors (a CPU tensor moves to the current device), gloo host tensors
    work = local.to(ctl_dev) if (nccl != local.is_cuda) else local
    sizes = [int(h[3]) for h in hdrs]
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
    if not active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if active():
        dist.barrier()
