"""Randomised sweep of the SHARDED solve on one GPU (TEST TOOLING): two ranks under torch.distributed (gloo: host-staged all-reduce hook), each
with its landmark shard of the same random window (tools/fuzz_parity.py: cases); rank 0 also solves the whole window on its own and compares —
reduced normal equations of the first linearisation 1e-9, same accept / reject sequence, 4-iteration end points 1e-6 (control points; each rank's own
landmarks). Windows with an IMU factor their bordered system from both ends with the border sweep next to the factorisation on every rank.
usage (GPU box): python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/fuzz_shards.py [cases=40] [seed=1] [large]"""
import os
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np


def main():
    import torch
    import torch.distributed as dist
    import hyperslam_amd as ha
    from hyperslam_amd import _lib, synthetic
    from hyperslam_amd.distributed import attach_allreduce
    from fuzz_parity import cases, rel
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    lib = _lib.load()
    referee = _lib.Library(os.path.join("oracle", "liboracle_ld.so"), "hsl_", strict=False)
    failures = 0
    for tag, w in cases(n_cases, seed, large):
        flag = torch.zeros(1, dtype=torch.int64)
        line = ""
        try:
            shard = synthetic.shard_by_landmark(w, rank, world)
            with ha.Problem(shard, lib=lib) as p:
                attach_allreduce(p, dist)
                S, g = p.reduced_system(1e4)
                s = p.solve(4)
                cp, lm = p.control_points(), p.landmarks()
            if rank == 0:
                with ha.Problem(w, lib=lib) as q:
                    bw = q.lib.band_blocks(q.h)
                    S1, g1 = q.reduced_system(1e4)
                    s1 = q.solve(4)
                    cp1, lm1 = q.control_points(), q.landmarks()
                ids = np.unique(np.concatenate([shard.pixel_landmark, shard.bearing_landmark])).astype(int)
                n = min(S.shape[0], S1.shape[0])
                errs = dict(S=rel(S[:n, :n], S1[:n, :n]), g=rel(g[:n], g1[:n]), final=abs(s["final_cost"] - s1["final_cost"]) / abs(s1["final_cost"]), cp=rel(cp, cp1),
                            lm=rel(lm[ids], lm1[ids]) if len(ids) else 0.0)
                same = [i["step_is_successful"] for i in s["iterations"]] == [i["step_is_successful"] for i in s1["iterations"]]
                ok = errs["S"] < 1e-9 and errs["g"] < 1e-9 and same and errs["final"] < 1e-6 and errs["cp"] < 1e-6 and errs["lm"] < 1e-6
                if not ok and errs["S"] < 1e-9 and errs["g"] < 1e-9 and same:
                    # same normal equations and decisions, end points apart: an ill-conditioned window (tools/fuzz_parity.py) — the long-double oracle
                    # referees: the sharded solve must be within an order of magnitude of the single-process solve's distance from it (+ 1e-7)
                    with ha.Problem(w, lib=referee) as r:
                        r.solve(4)
                        cpr = r.control_points()
                    errs["cp_shard_ld"], errs["cp_single_ld"] = rel(cp, cpr), rel(cp1, cpr)
                    ok = errs["cp_shard_ld"] <= 10.0 * errs["cp_single_ld"] + 1e-7
                line = f"{tag} bw {bw:2d} | " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()) + ("" if ok else "  <-- FAIL")
                flag[0] = 0 if ok else 1
        except Exception as e:
            msg = str(e)
            refused = "INVALID" in msg or "invalid" in msg or "too long" in msg or "span too many" in msg or "bad shard" in msg or "window refused" in msg  # (limits of DESIGN 8)
            line = f"{tag} | {'refused' if refused else 'ERROR'}: {msg[:120]}"
            flag[0] = 0 if refused else 1
        dist.all_reduce(flag)
        if rank == 0:
            print(line, flush=True)
        failures += int(flag.item() > 0)
    if rank == 0:
        print(f"{n_cases} cases on {world} ranks, {failures} failures")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(min(failures, 100))


if __name__ == "__main__":
    main()
