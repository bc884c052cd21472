"""Worker: the RCCL (backend "nccl") branch of the exchange hook on device memory, with a one-rank group (the test box has a
single GPU). Exercises the zero-copy tensor view of the library's buffer, the external-stream handoff and the collective."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import hyperslam_amd as ha
    from hyperslam_amd import synthetic
    from hyperslam_amd.distributed import attach_allreduce
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    w = synthetic.small_visual(order=4, n_cp=18, n_landmarks=64, obs_pairs=3, with_priors=21)
    with ha.Problem(w) as p:
        s0 = p.solve(5)
        cp0 = p.control_points()
    with ha.Problem(w) as p:
        attach_allreduce(p, dist)
        s1 = p.solve(5)
        cp1 = p.control_points()
    from hyperslam_amd.distributed import attach_rccl
    with ha.Problem(w) as p:
        assert attach_rccl(p, dist)  # library-owned communicator, ncclAllReduce on the library's stream
        s2 = p.solve(5)
        cp2 = p.control_points()
    np.savez(out, c0=[it["cost"] for it in s0["iterations"]], c1=[it["cost"] for it in s1["iterations"]], cp0=cp0, cp1=cp1,
             c2=[it["cost"] for it in s2["iterations"]], cp2=cp2)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
