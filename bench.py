#!/usr/bin/env python3
"""bench.py -- CHOMP iterations/sec on MI355X for the BASELINE.json workload.

  python bench.py --gpus N --steps K --warmup W

A *step* is one `iterate` of the hot path over one batch: n_iter=100 CHOMP
iterations (costs every iteration, final cost pass included) of `--batch` WAM-7
runs with 100 waypoints each (BASELINE.json configs[1]: WAM 7-DOF, n_points=100,
batch=1024 random adofgoal, fp64).  Every step works on its own freshly seeded
batch (created before the timed region), so all steps do identical work; the
trajectories are resident in HBM when the timed region starts.  For N > 1 every
rank owns `--batch` runs with its own goals (weak scaling, no data-path
collective: the runs are independent).

Rank 0 prints ONE JSON line; see the task contract for the fields.  `roofline`
prices the fused iterate kernel against HBM with the ALGORITHMIC bytes of
SURVEY.md 8(d) (58 040 B per iteration per run for this workload); the kernel
duration is measured live with HIP events on the stream the kernel is launched
on.  The steps are independent batches and are issued round-robin on `--streams`
HIP streams (default 3) so that the tail of one launch overlaps the next; the
per-launch duration (and with it `roofline.achieved`) is that of a launch that
shares the GPU with its neighbour.  `cpu_baseline` times the oracle (oracle/, a CPU restatement of the reference)
on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_POINTS = 100
N_ITER = 100
LAMBDA = 100.0
OBS_FACTOR = 500.0
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the FP32 vector rate of that guide (157.3 TF): a wave64 fp64 instruction holds a SIMD for 4 cycles
FLOP_PER_ITERATION = 0.8e6      # SURVEY.md 8(d) "Algorithmic flops", config W


def algorithmic_bytes_per_iter(m, n, Sa, n_sdf, w, momentum):
    """2 m n w (T) [+ 2 m n w momentum] + m Sa Nsdf 4 w (SDF gathers) + 3 w (costs)."""
    return 2 * m * n * w + (2 * m * n * w if momentum else 0) + m * Sa * n_sdf * 4 * w + 3 * w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="runs per GPU (configs[1]: 1024)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued on round-robin: consecutive steps are independent batches, "
                         "so the tail of one launch (a few slow runs) is filled by the next; 1 = strictly serial launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=0, help="override the cpu baseline sample size")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for a "
                                                      "single-GPU rehearsal of the multi-rank path)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=args.backend)

    import or_cdchomp_amd
    import common

    mod = or_cdchomp_amd.Module(device)
    model = common.setup_product_wam(mod)
    if args.streams > 1:
        mod.set_num_streams(args.streams)
    n_runs = args.batch
    kw = dict(n_points=N_POINTS, lambda_=LAMBDA, obs_factor=OBS_FACTOR)

    def make_batches(count, seed0):
        ids = []
        for k in range(count):
            goals = common.wam_goals(n_runs, seed=seed0 + 1000 * rank + k)
            ids.append(mod.batch_create(model.name, goals, **kw))
        return ids

    warm = make_batches(args.warmup, 30250101)
    timed = make_batches(args.steps, 20250101)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for bid in warm:
        mod.batch_iterate_async(bid, N_ITER)
    for bid in warm:
        mod.batch_sync(bid)
    mod.kernel_time(reset=True)

    barrier()
    t0 = time.perf_counter()
    for bid in timed:
        mod.batch_iterate_async(bid, N_ITER)
    results = [mod.batch_sync(bid) for bid in timed]      # costs + status come back with iterate
    barrier()
    t1 = time.perf_counter()
    from or_cdchomp_amd import sharding
    elapsed = sharding.max_over_ranks(t1 - t0, dist)

    kernel_ms, launches = mod.kernel_time()
    # host-side gather of the per-run verdicts of all ranks (untimed; gloo, no RCCL data path)
    local = {"status": np.concatenate([st for _, st in results]), "costs": np.concatenate([c for c, _ in results])}
    whole = sharding.gather_host(local, dist, sharding.host_group(dist))
    status_bad = int((whole["status"] != 0).sum()) if whole is not None else 0

    # ---- parity spot check against the oracle on the first timed batch (untimed) ----
    parity = None
    parity_ill = []
    k_check = 0
    cpu = None
    if rank == 0:
        from oracle import oracle_py as O
        O.build(ref=False)
        prob = common.tabletop_problem(O)
        rob = O.OraRobot(model)
        _, base, dofvals, adofs = common.wam_state()
        p = O.default_params(**kw)
        goals0 = common.wam_goals(n_runs, seed=20250101)
        traj0 = mod.batch_gettraj(timed[0])
        k_check = min(16, n_runs)
        otraj, ocosts, ost, _ = O.batch_run(rob, base, dofvals, adofs, goals0[:k_check], [prob["sdf"]],
                                            [prob["pose"]], p, N_ITER, max_threads=k_check)
        # Some runs of this workload are chaotic in the reference algorithm itself (momentum-free
        # CHOMP bouncing off joint limits amplifies rounding x4 per projection):
        # the oracle run again with the goals moved by ONE ulp shows which, and how far such a run
        # may legitimately drift.  Parity is quoted over the well-conditioned runs; the others are
        # listed with both figures.
        ptraj, _, _, _ = O.batch_run(rob, base, dofvals, adofs, goals0[:k_check] * (1.0 + 2.0 ** -52), [prob["sdf"]],
                                     [prob["pose"]], p, N_ITER, max_threads=k_check)
        errs = [common.rel_l2(traj0[k], otraj[k]) for k in range(k_check)]
        self_amp = [common.rel_l2(ptraj[k], otraj[k]) for k in range(k_check)]
        well = [k for k in range(k_check) if self_amp[k] < 1e-9 and ost[k] == 0]
        parity = max(errs[k] for k in well) if well else None
        parity_ill = [{"run": k, "hip_vs_oracle": errs[k], "oracle_vs_oracle_goal_plus_one_ulp": self_amp[k]}
                      for k in range(k_check) if k not in well]

        if not args.no_cpu_baseline and world == 1:         # the CPU baseline is reported at N=1 only
            cores = os.cpu_count() or 1
            try:                                  # the container's CPU quota (cgroup v2), when there is one
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    cores = max(1, min(cores, int(round(int(quota) / int(period)))))
            except (OSError, ValueError):
                pass
            cores = min(cores, len(os.sched_getaffinity(0)))
            # ~0.1 s per run (100 iterations) on one core; aim for 10-20 s of wall time
            sample = args.cpu_runs or int(min(n_runs, max(8, 48 * cores)))
            sample = min(sample, n_runs)
            c0 = time.perf_counter()
            _, _, _, threads = O.batch_run(rob, base, dofvals, adofs, goals0[:sample], [prob["sdf"]],
                                           [prob["pose"]], p, N_ITER, max_threads=cores)
            c1 = time.perf_counter()
            # the same code on ONE core (SURVEY.md 8d asks for both): 8 runs
            one = min(8, n_runs)
            s0 = time.perf_counter()
            O.batch_run(rob, base, dofvals, adofs, goals0[:one], [prob["sdf"]], [prob["pose"]], p, N_ITER, max_threads=1)
            s1 = time.perf_counter()
            cpu = {"value": sample * N_ITER / (c1 - c0), "unit": "CHOMP iterations/s", "cores": int(threads),
                   "value_1_core": one * N_ITER / (s1 - s0), "host_cores": int(cores),
                   "kind": "port",
                   "sample": "%d of the %d runs of step 0 x %d iterations, oracle (C restatement of libcd + "
                             "sphere cost, dense A^-1 as the reference), OpenMP over runs, %.1f s wall"
                             % (sample, n_runs, N_ITER, c1 - c0)}

    if rank == 0:
        total_iters = float(world) * n_runs * N_ITER * args.steps
        value = total_iters / elapsed
        m, n, Sa = N_POINTS - 2, 7, 15
        bytes_iter = algorithmic_bytes_per_iter(m, n, Sa, 1, 8, False)
        avg_ms = kernel_ms / max(launches, 1)
        bytes_launch = bytes_iter * n_runs * N_ITER
        achieved = bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("batch") == n_runs and tj.get("n_iter") == N_ITER:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "CHOMP iters/sec, 7-DOF x 100-waypoint",
            "value": value,
            "unit": "CHOMP iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "WAM 7-DOF, n_points=100, batch=%d random adofgoal per GPU, n_iter=%d per step, "
                                   "lambda=100 obs_factor=500, tabletop SDF 40x31x12 (BASELINE configs[1])"
                                   % (n_runs, N_ITER),
                       "runs_per_gpu": n_runs, "n_iter": N_ITER, "n_points": N_POINTS, "dof": 7,
                       "parallelism": "runs sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "chomp_iterate_kernel<double>", "avg_kernel_ms": avg_ms, "launches": launches,
                         "concurrent_launches": max(1, args.streams),
                         # all overlapping launches together: algorithmic bytes of the K steps / their wall time
                         "aggregate_achieved": value / world * bytes_iter / 1e9,
                         "aggregate_frac": value / world * bytes_iter / 1e9 / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "algorithmic_bytes_per_iteration_per_run": bytes_iter,
                         # the kernel is bound by the fp64 vector pipe, not by HBM:
                         # SURVEY.md 8(d) asks for this figure beside the HBM one
                         "fp64_vector": {"achieved": value * FLOP_PER_ITERATION / 1e12 / world,
                                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": value * FLOP_PER_ITERATION / 1e12 / world / FP64_VECTOR_PEAK_TFLOPS,
                                         "algorithmic_flop_per_iteration_per_run": FLOP_PER_ITERATION}},
            "cpu_baseline": cpu,
            "parity_rel_l2_max_vs_oracle": parity,
            "parity_runs_checked": k_check, "parity_ill_conditioned_runs": parity_ill,
            "runs_outside_joint_limits": status_bad,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
