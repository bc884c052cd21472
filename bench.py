#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json: residuals/sec + ms/Gauss-Newton iter, 128-ctrl-pt window).

Step = one optimize() call (the reference's unit of work: Ceres trust-region LM with max_num_iterations = 5,
/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:40,276-280) on the BASELINE.json configs[1] window
(order-4 spline, 128 control points, 50 000 pixel-reprojection residual blocks, 5 000 landmarks), restarted from the same
HBM-resident window state every step (device-side restore; tables are uploaded before the timed region).
Every step executes exactly 5 LM iterations (asserted), each = linearise all residual blocks -> robustify -> landmark
Schur complement -> banded reduced solve -> retract -> cost re-evaluation -> accept/reject.
value = residual blocks linearised per second over the whole job (all ranks) = blocks * 5 * steps / time.

N > 1 (torchrun, one rank per GPU): residual blocks are sharded by landmark; every rank holds the
replicated control points, eliminates its own landmarks and the reduced normal equations are summed with one RCCL
all-reduce per iteration. Weak scaling: each rank gets a full configs[1]-sized shard (global = N x 50k residual blocks
observing N x 5k landmarks on the same 128 control points).

--config 3: BASELINE.json configs[3] (512 control points, 200 k residual blocks, 20 k landmarks) sharded by landmark over the N ranks,
STRONG scaling (total work fixed). --config 2: configs[2] (stereo-inertial, order 6) on one GPU. The default (configs[1]) is the
configuration the metric is quoted on.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LM_ITERATIONS = 5           # optimizer.cpp:40
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6     # MI355X_MICROARCH.md: vector fp64 = half the 157.3 TFLOP/s fp32 rate (the f64 MFMA runs at the same rate, tools/microbench)


def cpu_baseline(window, budget_s=20.0):
    """Oracle (CPU restatement, 1 thread like the reference's num_threads = 1) timed on the same workload, bounded."""
    import hyperslam_amd as ha
    from hyperslam_amd import _lib
    # The shipped liboracle.so is built for a portable target (x86-64-v3) because it travels between machines; the timed baseline is
    # compiled like the reference (-O3 -march=native, CMakeLists.txt:23) on the host that runs it, falling back to the shipped one.
    import subprocess
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            cpu = "".join(l for l in f.read().split("\n\n")[0].splitlines(True) if l.startswith(("model name", "flags")))
    except OSError:
        cpu = "unknown"
    native = os.path.join(ROOT, "oracle", f"liboracle_native_{hashlib.sha1(cpu.encode()).hexdigest()[:10]}.so")  # per host CPU: in-tree .so files travel
    kind_note = "-O3 -march=native build on this host"
    try:
        src = os.path.join(ROOT, "oracle", "capi.cpp")
        import glob
        newest = max(os.path.getmtime(f) for f in [src] + glob.glob(os.path.join(ROOT, "oracle", "*.hpp")))
        if not os.path.exists(native) or os.path.getmtime(native) < newest:
            subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", "-fPIC", "-shared", "-o", native, src],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        lib = _lib.Library(native, "hso_")
    except Exception:
        lib = _lib.Library(os.path.join(ROOT, "oracle", "liboracle.so"), "hso_")
        kind_note = "portable x86-64-v3 build"
    n_blocks = window.num_residual_blocks()
    runs, spent, iters = 0, 0.0, 0
    while runs < 1 or (spent < budget_s and runs < 5):
        with ha.Problem(window, lib=lib) as p:
            t = time.perf_counter()
            s = p.solve(LM_ITERATIONS)
            spent += time.perf_counter() - t
        iters += s["num_iterations"]
        runs += 1
    out = {"value": n_blocks * iters / spent, "unit": "residual_blocks/s", "cores": 1, "kind": "port",
           "ms_per_iteration": 1e3 * spent / iters,
           # SURVEY.md §8d per-stage CPU times (last run): linearise / Schur build / reduced solve / retract + cost re-evaluation
           "stage_ms_per_iteration": {k: s[k] / max(1, s["num_iterations"]) for k in ("linearize_ms", "schur_ms", "solve_ms", "update_ms")},
           "sample": f"{runs} x optimize() ({LM_ITERATIONS} LM iterations) of the full workload on 1 host thread; own C++ restatement, not Ceres; {kind_note}"}
    out["all_cores"] = cpu_all_cores(lib.path, n_blocks)
    return out


def cpu_all_cores(lib_path, n_blocks):
    """SURVEY.md §8d's second CPU figure. The reference solves one window on one thread (optimizer.cpp:41), so the only way it fills
    a host is with independent windows: one optimize() of the same workload per hardware thread, each in its own PROCESS (threads
    of one process serialise on the address-space lock while they fault in their working sets), all started at the same wall-clock
    instant; value = blocks linearised by all processes / (latest finish - common start)."""
    import subprocess
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    start = time.time() + 8.0 + 0.04 * cores  # interpreter start-up + window generation of every worker
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", lib_path, repr(start)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
    results = []
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        if pr.returncode == 0 and out.strip():
            results.append(json.loads(out.strip().splitlines()[-1]))
    if not results:
        return None
    wall = max(r["end"] for r in results) - start
    return {"value": n_blocks * sum(r["iters"] for r in results) / wall, "unit": "residual_blocks/s", "cores": len(results),
            "sample": f"{len(results)} independent windows (one optimize() = {LM_ITERATIONS} LM iterations each, one process per hardware thread) "
                      f"started together, {wall:.2f} s wall; {sum(r['late'] for r in results)} started late"}


def cpu_worker(lib_path, start):
    """One process of the all-cores CPU leg (bench.py --cpu-worker): the oracle on the configs[1] window, started at `start`."""
    import hyperslam_amd as ha
    from hyperslam_amd import _lib, synthetic
    lib = _lib.Library(lib_path, "hso_")
    problem = ha.Problem(synthetic.config1(n_cp=128, n_landmarks=5000, obs_pairs=5), lib=lib)
    late = time.time() > start
    while time.time() < start:
        time.sleep(0.0005)
    s = problem.solve(LM_ITERATIONS)
    print(json.dumps({"end": time.time(), "iters": s["num_iterations"], "late": int(late)}), flush=True)


def _newest(pattern):
    """Committed profile files in natural order (r01_v11 after r01_v7)."""
    import glob
    import re
    return sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), or None."""
    files = _newest("r*_pmc_hbm_traffic.json")
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"][kernel]
        return 1024.0 * (k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"])
    except Exception:
        return None


def n_cp_rows(window):
    return int(window.control_points.shape[0])


def rocprof_kernel_ms(kernel_prefix):
    """Average duration of the kernel in the committed rocprofv3 --kernel-trace --stats summary of this command, or None."""
    import csv
    files = _newest("r*_bench_kernel_stats.csv")
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if row["Name"].startswith(kernel_prefix):
                    return float(row["AverageNs"]) * 1e-6
    except Exception:
        pass
    return None


def dominant_kernel(np_rows, bw, two_ended):
    """The kernel with the largest share of device time in the newest committed rocprofv3 kernel-trace summary of this command
    (profiles/r*_bench_kernel_stats.csv), priced against both rooflines with an algorithmic model of the banded factorisation:
      bytes  = band of S read + band of U written + right-hand side in/out          (8 B x (2 np ncb + 2 np), ncb = 6 bw)
      flops  = per block row: 6x6 Cholesky + 6 x ncb panel solve + symmetric rank-6 update of the trailing band  ~ 6 ncb^2 + 72 ncb
    Both fractions are tiny by construction: the factorisation is ONE dependency chain of np / 6 block rows on one or two workgroups
    (DESIGN.md §6); the entry exists so that the share of the iteration it takes is not hidden behind the linearisation's roofline."""
    import csv
    files = _newest("r*_bench_kernel_stats.csv")
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rows = [r for r in csv.DictReader(f) if "hs::" in r["Name"]]
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    except Exception:
        return None
    avg_s = float(top["AverageNs"]) * 1e-9
    out = {"kernel": top["Name"].replace("void ", "").split("(")[0], "share_of_device_time": float(top["TotalDurationNs"]) / total,
           "avg_kernel_ms": avg_s * 1e3, "calls": int(top["Calls"]), "source": os.path.relpath(files[-1], ROOT)}
    if "k_band_factor" in top["Name"]:
        ncb, n_blk = 6 * bw, np_rows // 6
        nbytes = 8.0 * (2 * np_rows * ncb + 2 * np_rows)
        flops = n_blk * (6.0 * ncb * ncb + 72.0 * ncb)
        wgs = 2 if two_ended else 1
        threads = (7 * 64 if ", 4>" in top["Name"] else 6 * 64) if "_la<" in top["Name"] else (9 * 64 if "_mx" in top["Name"] else 6 * 64 if "mfma" in top["Name"] else 256)
        out.update({"algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
                    "hbm": {"achieved": nbytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS},
                    "fp64": {"achieved": flops / avg_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / FP64_PEAK_TFLOPS},
                    "workgroups": wgs, "waves_launched": wgs * threads // 64, "waves_available": 256 * 4 * 8,
                    "bound": "latency (single dependency chain over the block rows: ~0.95 us per block row on k_band_factor_mx — trailing update on "
                             "v_mfma_f64_16x16x4_f64, panel on the vector unit — 1.2 us on k_band_factor_la, DESIGN.md §5)"})
        if "_mx" in top["Name"]:  # matrix-core share of the same launch: 2 MFMAs (K = 6 padded to 8) per active 16 x 16 tile and block row
            n_mfma = int(n_blk * 2 * 17.25)  # (every block row of the system is applied once, on one end or the other)
            out["mfma_f64"] = {"instructions_per_launch": n_mfma, "flops_issued": n_mfma * 2.0 * 16 * 16 * 4,
                               "note": "21 ring tiles per end, 15 or 21 of them inside the trailing band per phase (17.25 on average); the useful part of a "
                                       "16x16x4 tile update is 6/8 of its K and ~56 % of its area (72 x 72 upper triangle in 21 tiles)"}
    return out


def bearing_variant(ha, synthetic, device, steps=10):
    """configs[1] with bearing (AngularMetric) instead of pixel residuals — not part of `value`."""
    import torch
    w = synthetic.config1(n_cp=128, n_landmarks=5000, obs_pairs=5, bearing=True)
    with ha.Problem(w, device=device) as p:
        p.snapshot()
        for _ in range(3):
            p.restore(), p.solve(LM_ITERATIONS)
        torch.cuda.synchronize()
        t0, iters = time.perf_counter(), 0
        for _ in range(steps):
            p.restore()
            iters += p.solve(LM_ITERATIONS)["num_iterations"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"workload": "configs[1] with bearing residual blocks (VisualBearingEvaluator + AngularMetric, Huber 1.6e-3)", "steps": steps,
            "ms_per_gn_iteration": 1e3 * dt / iters, "value": w.num_residual_blocks() * iters / dt, "unit": "residual_blocks/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3], help="BASELINE.json configs[i]; 1 = the metric's configuration (default)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", nargs=2, metavar=("LIB", "START"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker[0], float(args.cpu_worker[1]))

    import numpy as np
    import torch
    import hyperslam_amd as ha
    from hyperslam_amd import synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if os.environ.get("HS_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0  # all ranks share GPU 0 (path test only)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("HS_DIST_BACKEND", "nccl")  # "gloo" only to exercise the N > 1 path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.config == 1:  # weak scaling: world x configs[1]; landmarks l with l % world == rank live on this rank
        full = synthetic.config1(n_cp=128, n_landmarks=5000 * world, obs_pairs=5)
        scaling, workload = "weak", ("BASELINE.json configs[1]: order-4 SE3 B-spline, 128 control points, 50k pixel reprojection residual blocks "
                                     "+ 5k landmarks per GPU, Schur on landmarks")
    elif args.config == 3:  # strong scaling: the one configs[3] window, its landmarks dealt over the ranks
        full = synthetic.config3()
        scaling, workload = "strong", ("BASELINE.json configs[3]: order-4 SE3 B-spline, 512 control points, 200k pixel residual blocks + 20k landmarks "
                                       "in total, sharded by landmark over the ranks")
    else:
        if world > 1:
            raise SystemExit("--config 2 (stereo-inertial) is a single-GPU configuration")
        full = synthetic.config2()
        scaling, workload = "weak", "BASELINE.json configs[2]: order-6 SE3 B-spline, 128 control points, 50k pixel + 10k inertial residual blocks, bias splines + gravity"
    window = synthetic.shard_by_landmark(full, rank, world) if world > 1 else full
    n_blocks_local = window.num_residual_blocks()
    n_blocks_global = full.num_residual_blocks()

    from hyperslam_amd.distributed import attach_allreduce, attach_rccl
    problem = ha.Problem(window, device=local_rank)
    keep = None
    exchange = "rccl"
    if world > 1:
        if dist.get_backend() == "nccl" and os.environ.get("HS_EXCHANGE", "rccl") == "rccl":
            if not attach_rccl(problem, dist):  # ncclAllReduce enqueued by the library on its own stream
                keep = attach_allreduce(problem, dist)  # agreed fallback: torch.distributed all_reduce on the library's stream
                exchange = "hook"
        else:
            exchange = "hook"
            keep = attach_allreduce(problem, dist)  # Python hook (gloo test path / HS_EXCHANGE=hook)
    problem.snapshot()

    # Per-stage HIP events (4 per LM iteration, each a barrier packet worth ~5.7 us of idle device) are off in the library by default.
    # One step in STAGE_EVERY of the timed region runs with them on: the live launch duration of the linearisation kernel (roofline) and
    # the stage breakdown come from those launches, inside the timed region, at an eighth of the events' cost to `value`
    # (every 4th step until the end of round 4: 2.5 % of `value` went to the instrumentation; 25 sampled launches of 200 are plenty).
    STAGE_EVERY = 8

    def step(i=0):
        staged = i % STAGE_EVERY == 0
        problem.set_stage_timing(staged)
        problem.restore()
        s = problem.solve(LM_ITERATIONS)
        assert s["num_iterations"] == LM_ITERATIONS, s
        return s, staged

    for i in range(args.warmup):
        step(i)
    stage = {"linearize_ms": 0.0, "schur_ms": 0.0, "solve_ms": 0.0, "update_ms": 0.0, "total_ms": 0.0}
    n_staged = 0
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        s, staged = step(i)
        if staged:
            n_staged += 1
            for k in stage:
                stage[k] += s[k]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what the library itself says about the exchange: ranks of its RCCL communicator (ncclCommCount; 0 = hook / single shard) and the bytes
    # every rank contributes per LM iteration (one all-reduce of the reduced normal equations + one of five doubles for the decision)
    import ctypes
    n_ranks, n_lin, n_dec = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
    problem._check(problem.lib.exchange_info(problem.h, ctypes.byref(n_ranks), ctypes.byref(n_lin), ctypes.byref(n_dec)), "exchange_info")
    exchange_info = {"rccl_ranks": int(n_ranks.value), "exchange_bytes_per_iteration_per_rank": 8 * int(n_lin.value + n_dec.value) if world > 1 else 0}
    if world > 1 and exchange == "rccl":
        assert n_ranks.value == world, (n_ranks.value, world)  # the library's communicator spans every rank of the job
    if rank == 0:
        # launches of the linearise kernel bracketed by HIP events (every STAGE_EVERY-th step of the timed region). A solve of N iterations
        # launches it N times: the start point + the candidates of iterations 0 .. N - 2 on the speculative path (the last iteration only
        # costs its candidate), one per iteration otherwise; the library books all of them under linearize_ms (hyperslam_hip.h).
        n_lin = n_staged * LM_ITERATIONS
        lin_ms = stage["linearize_ms"] / n_lin
        order = int(window.order)
        n_visual = len(window.pixel_stamps) + len(window.bearing_stamps)
        bw_blocks = problem.lib.band_blocks(problem.h)
        # Default path (round 4): the FUSED BUILD — k_build_visual linearises a landmark group's residual blocks into LDS, eliminates its landmarks and
        # accumulates J_p'J_p - Yh Yh' in the same pass; the 448-byte record of SURVEY.md 8(d) (B_out at k = 4) is never written. Its
        # algorithmic bytes are what the kernel's contract makes it move: 32 B of inputs per residual block, the Y-hat rows (3 x 6 doubles
        # per control point a landmark touches: k_backsub_retract needs them), 25 doubles of per-landmark factors and one window partial
        # [bw (bw + 1) / 2 tiles of 36 + 3 x 6 bw doubles] per chunk of <= 24 landmarks. The record path (HS_BUILD_PATH=records, long feature
        # tracks) keeps the round-3 kernel and SURVEY's B_alg = 480 B per block.
        fused = os.environ.get("HS_BUILD_PATH") != "records" and bw_blocks * (bw_blocks + 1) // 2 <= 256 and n_visual > 0
        if fused:
            lin_kernel = f"hs::k_build_visual<{order}>"
            lm_ids = np.concatenate([window.pixel_landmark, window.bearing_landmark])
            st_all = np.concatenate([window.pixel_stamps, window.bearing_stamps])
            seg = np.floor((st_all - window.t0) / window.dt).astype(np.int64) - (order - 1) // 2
            lo = np.full(len(window.landmarks), 1 << 30, np.int64)
            hi = np.full(len(window.landmarks), -1, np.int64)
            np.minimum.at(lo, lm_ids, seg), np.maximum.at(hi, lm_ids, seg + order - 1)
            ncp_l = (hi - lo + 1)[hi >= 0]
            n_chunks = max(1, int(np.ceil(len(ncp_l) / 24.0)))  # lower bound of the chunk count (the library splits per landmark group)
            alg_bytes = 32 * n_visual + 8 * (18 * int(ncp_l.sum()) + 25 * len(ncp_l)) + 8 * n_chunks * (36 * bw_blocks * (bw_blocks + 1) // 2 + 18 * bw_blocks)
            # fp64 work of the same launch (the kernel is instruction-issue bound, not HBM bound: DESIGN.md §5): per residual block the
            # linearisation (~3.5 kFLOP: spline, factor, local Jacobian, loss), H_ll / b_l / the k W blocks (36 + 72 k) and k (k + 1) / 2
            # J_p'J_p tiles of 2 x 36 FMAs; per landmark n (n + 1) / 2 window tiles of 3 x 36 FMAs, Y-hat (108 n) and the 3 x 3 factor
            alg_flops = n_visual * (3500 + 36 + 72 * order + 144 * order * (order + 1) // 2) + int((ncp_l * (ncp_l + 1) // 2 * 216 + 108 * ncp_l + 100).sum())
            alg_note = ("32 B in per residual block + Y-hat rows + per-landmark factors + one window partial per chunk (lower bound: ceil(landmarks / 24) chunks); "
                        "the 448-byte record of SURVEY.md 8(d) is not materialised on this path")
        else:
            alg_flops = None
            lin_kernel = f"hs::k_linearize_visual<{order}>"
            alg_bytes = (32 + 8 * (8 + 12 * order)) * n_visual  # 32 B in + one record [r(2) J_l(6) J_state(12 k)] out = 480 B at k = 4
            alg_note = "SURVEY.md 8(d): B_alg = 32 B in + 8 (8 + 12 k) B record out per residual block"
        b_alg = alg_bytes / max(1, n_visual)
        live = alg_bytes / (lin_ms * 1e-3) / 1e9
        # the duration in the committed rocprofv3 kernel trace of this command excludes the ~6 us of dispatch latency the HIP events around
        # the launch include: it prices the kernel, the events price the launch. frac uses the trace when one is committed for this kernel.
        prof_ms = rocprof_kernel_ms(f"void {lin_kernel}") if args.config == 1 and world == 1 else None
        profiled = alg_bytes / (prof_ms * 1e-3) / 1e9 if prof_ms else None
        # `achieved` / `frac` are the numbers measured in THIS run (HIP events around the launch on the library's stream, inside the timed
        # region); the figures derived from the committed rocprofv3 trace of the same command stand beside them (`*_profile`).
        # SURVEY.md 8(d)'s own figure over the same launch time, beside the builder's byte model (the contract number: 480 B per pixel block at
        # k = 4 = 32 B in + the 448-byte record, + the shared tables once): what the kernel would be credited with if it materialised the record
        survey_bytes = (32 + 8 * (8 + 12 * order)) * n_visual + 64 * n_cp_rows(window) + 24 * len(window.landmarks)
        t_best_ms = prof_ms or lin_ms
        roofline = {"kernel": lin_kernel, "bound": "hbm", "achieved": live, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": live / HBM_PEAK_GBS,
                    "survey_bytes_per_launch": survey_bytes, "achieved_survey_bytes": survey_bytes / (t_best_ms * 1e-3) / 1e9,
                    "frac_survey_bytes": survey_bytes / (t_best_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": pmc_traffic(lin_kernel) if args.config == 1 and world == 1 else None,
                    "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_residual_block": b_alg, "algorithmic_bytes_model": alg_note,
                    "avg_launch_ms": lin_ms, "timing_source": "HIP events (live, this run)",
                    "rocprof_avg_kernel_ms": prof_ms, "achieved_profile": profiled, "frac_profile": profiled / HBM_PEAK_GBS if profiled else None,
                    "note": "avg_launch_ms / achieved / frac = HIP events around the launch on the library's stream, measured in this run (they include "
                            "~4 us of dispatch latency); rocprof_avg_kernel_ms / achieved_profile / frac_profile = committed rocprofv3 --kernel-trace "
                            "--stats average of this command (the kernel alone); traffic = FETCH_SIZE + WRITE_SIZE of the newest "
                            "profiles/r*_pmc_hbm_traffic.json (separate rocprofv3 --pmc passes, tools/pmc_traffic.sh)"}
        if alg_flops:
            t_ms = prof_ms or lin_ms
            roofline["fp64_valu"] = {"achieved": alg_flops / (t_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": alg_flops / (t_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": alg_flops,
                                     "time_ms": t_ms, "note": "vector fp64 (no MFMA: 6-wide blocks, DESIGN.md §2); the kernel's binding resource"}
        n_cp_total = int(window.control_points.shape[0])
        # (launch_factor's rule: look-ahead kernel, window of >= 4 bw block rows; bordered systems too unless HS_DEBUG_FLAGS 536870912 / 2048)
        flags = int(os.environ.get("HS_DEBUG_FLAGS", "0"))
        two_ended = n_cp_total >= 4 * bw_blocks and bw_blocks <= 16 and not (flags & 2048) and not (len(window.inertial_stamps) and (flags & 536870912))
        out = {
            "metric": "residual blocks linearised per second (LM iteration = linearise + Schur + solve + update), 128-control-point window",
            "value": n_blocks_global * LM_ITERATIONS * args.steps / elapsed,
            "unit": "residual_blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_gn_iteration": 1e3 * elapsed / (args.steps * LM_ITERATIONS),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "residual_blocks_per_gpu": n_blocks_local, "residual_blocks_total": n_blocks_global, "landmarks_per_gpu": int(len(np.unique(np.concatenate([window.pixel_landmark, window.bearing_landmark])))),
                       "lm_iterations_per_step": LM_ITERATIONS, "parallelism": f"residual-sharded x{world}" if world > 1 else "single GPU",
                       **({"exchange": exchange} if world > 1 else {}), **exchange_info},
            "final_cost": s["final_cost"], "initial_cost": s["initial_cost"],
            "device_ms_per_iteration": {k: v / n_lin for k, v in stage.items()},
            "stage_events": f"HIP events around the four stages on every {STAGE_EVERY}th step of the timed region ({n_staged} of {args.steps} steps); "
                            "those steps carry ~23 us of event barriers per iteration, the others none",
            # roofline of the kernel SURVEY.md §8(d)'s B_alg is defined for (linearisation: 480 B per pixel residual block); its
            # launch time is measured with HIP events on the launch stream inside hs_solve. The factorisation kernel that
            # dominates the iteration time is a single-workgroup dependency chain (latency-bound, no meaningful roofline); its
            # share is visible in device_ms_per_iteration["solve_ms"].
            "roofline": roofline,
        }
        # (the committed kernel trace belongs to the default command: configs[1] on one GPU)
        dom = dominant_kernel(6 * n_cp_total, bw_blocks, two_ended) if args.config == 1 and world == 1 else None
        if dom:
            out["roofline_dominant"] = dom
        # the same algorithmic bytes against the whole LM iteration (linearise + Schur + solve + update): the path is a latency-bound
        # dependency chain after the linearisation, so this fraction is low by construction
        it_ms = out["ms_per_gn_iteration"]
        out["roofline_iteration"] = {"bound": "hbm", "achieved": alg_bytes / (it_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": alg_bytes / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms": it_ms}
        if args.config == 1 and world == 1:  # the factor the runtime actually instantiates (abstract.cpp:243-260): same window, bearing residuals
            out["bearing_variant"] = bearing_variant(ha, synthetic, local_rank)
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (the other ranks would idle behind rank 0's CPU run)
            out["cpu_baseline"] = cpu_baseline(full)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
            if out["cpu_baseline"].get("all_cores"):
                out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores"]["value"]
        print(json.dumps(out))
    problem.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
