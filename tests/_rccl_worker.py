"""Child process of tests/test_gpu_dist.py::test_rccl_branches_at_world_size_one: ONE rank, backend "nccl" (= RCCL on ROCm),
TDRN_DIST_FORCE_GROUP=1.  Drives every RCCL branch of tdrn_amd/dist.py and the weight broadcast of the model shell on cuda:0 --
the same ncclCommInitRank / ncclBroadcast / ncclAllGather / ncclGather (send-recv) / ncclAllReduce entry points an 8-rank job
takes, before the driver's first multi-GPU run does.  Started fresh: the process group is built before the first GPU call.
What a one-rank group cannot show: the padded (ragged) payload of gather_results and a peer's header mismatch -- those two
run over gloo at world 2 and 8 in tests/test_dist_cpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from tdrn_amd import dist as tdist
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth


def main():
    out_path = sys.argv[1]
    rank, local_rank, world = tdist.init()                      # nccl: chosen because a GPU is visible (no TDRN_DIST_BACKEND)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl" and tdist.active()
    dev = torch.device("cuda", local_rank)

    # the one collective of the path: rank 0 packs, the blob goes through ncclBroadcast, the engine adopts it
    def make():
        net = build_net("test", 320, 21, 1024, 1, True, True)
        net.set_compute_dtype("bf16")
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return net.eval()
    eng = make().adopt_broadcast_weights(src=0, device=dev)
    plain = make().engine(dev)                                   # the same weights packed without any collective
    x = torch.from_numpy(synth.synth_frames(2, 320, seed=51)).to(dev)
    a, b = eng.forward(x), plain.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(eng.weights, plain.weights)
    for k in ("arm_loc", "odm_loc", "conf"):
        assert torch.equal(a[k], b[k]), k
    # NetEngine.broadcast_weights on an engine that never packed: allocates, receives (its own zeros at world 1), adopts
    blob = torch.arange(1 << 20, dtype=torch.int32, device=dev)
    keep = blob.clone()
    assert tdist.broadcast_blob(blob, src=0) is blob and torch.equal(blob, keep)

    # gather_results: tensor on the device, tensor on the host (moved to the device for RCCL and back), object, 0-dim tensor
    det = torch.randn(3, 21, 200, 5, device=dev)
    got = tdist.gather_results(det, rank, world)
    assert len(got) == 1 and got[0].device == det.device and torch.equal(got[0], det)
    host = torch.arange(24, dtype=torch.int64).reshape(4, 6)
    got = tdist.gather_results(host, rank, world)
    assert len(got) == 1 and got[0].device.type == "cpu" and torch.equal(got[0], host)
    empty = torch.zeros(0, 21, 200, 5, device=dev)               # an empty shard keeps its trailing shape
    got = tdist.gather_results(empty, rank, world)
    assert got[0].shape == empty.shape
    assert tdist.gather_results({"frames": 7}, rank, world) == [{"frames": 7}]
    z = tdist.gather_results(torch.tensor(2.5, device=dev), rank, world)
    assert len(z) == 1 and float(z[0]) == 2.5
    for bad in (torch.zeros(2, 2, dtype=torch.complex64, device=dev), torch.zeros((1,) * 9, device=dev)):
        try:
            tdist.gather_results(bad, rank, world)
        except ValueError as e:
            assert "unsupported tensor" in str(e)
        else:
            raise AssertionError("an unsupported tensor must raise on every rank")

    assert tdist.max_over_ranks(3.25) == 3.25                    # device chosen from the backend
    assert tdist.max_over_ranks(1.5, dev) == 1.5
    tdist.barrier()
    torch.cuda.synchronize()
    np.savez(out_path, ok=np.asarray([1]), wsum=np.asarray([float(eng.weights.double().sum())]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
