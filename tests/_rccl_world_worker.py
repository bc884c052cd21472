"""Worker of test_rccl_two_ranks_on_two_gpus: rank r on GPU r solves its landmark shard with the library-owned RCCL communicator
(hs_rccl_unique_id on rank 0 -> broadcast -> hs_rccl_init(world) on every rank; ncclAllReduce on the library's stream inside hs_solve)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import hyperslam_amd as ha
    from hyperslam_amd import synthetic
    from hyperslam_amd.distributed import attach_rccl
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    full = synthetic.config3(n_cp=64, n_landmarks=600, obs_pairs=5)
    shard = synthetic.shard_by_landmark(full, rank, world)
    with ha.Problem(shard, device=rank) as p:
        assert attach_rccl(p, dist)
        S, g = p.reduced_system(1e4)
        s = p.solve(5)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), S=S, g=g, cp=p.control_points(), lm=p.landmarks(),
                 lm_ids=np.unique(np.concatenate([shard.pixel_landmark, shard.bearing_landmark])),
                 costs=np.array([it["cost"] for it in s["iterations"]]), iters=s["num_iterations"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
