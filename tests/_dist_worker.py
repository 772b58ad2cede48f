"""Child process of tests/test_gpu_dist.py (one rank of a frame-sharded job).  Started fresh -- nothing has touched the
GPU before torch.distributed is up -- with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment, like torchrun."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from tdrn_amd import dist as tdist
from tdrn_amd.model.dualrefinedet_vggbn import build_net
from tdrn_amd.utils import synth


def main():
    out_dir = sys.argv[1]
    rank, local_rank, world = tdist.init()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    net = build_net("test", 320, 21, 1024, 1, True, True)
    net.set_compute_dtype("bf16")
    if rank == 0:                                   # only rank 0 has the checkpoint; the others keep default-init modules
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    eng = net.adopt_broadcast_weights(src=0, device=dev)          # the one collective of the path
    frames = synth.synth_frames(4, 320, seed=51)                    # the same 4 frames everywhere; each rank also runs its shard
    x = torch.from_numpy(frames).to(dev)
    r = eng.forward(x)
    mine = tdist.shard_slice(4, rank, world)
    rs = eng.forward(x[mine])
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), arm=r["arm_loc"].cpu().numpy(), odm=r["odm_loc"].cpu().numpy(),
             conf=r["conf"].cpu().numpy(), shard_odm=rs["odm_loc"].cpu().numpy(), shard=np.asarray([mine.start, mine.stop]),
             wsum=np.asarray([float(eng.weights.double().sum())]))
    t = tdist.max_over_ranks(1.0 + rank, dev)
    assert t == float(world)
    gathered = tdist.gather_results(int(mine.start), rank, world)
    if rank == 0:
        assert gathered == [tdist.shard_slice(4, k, world).start for k in range(world)]
    tdist.barrier()


if __name__ == "__main__":
    main()
