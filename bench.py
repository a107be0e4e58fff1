#!/usr/bin/env python3
"""bench.py -- CHOMP iterations/sec on MI355X for the BASELINE.json workloads.

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

With N > 1 and no WORLD_SIZE in the environment the command starts its own N ranks (`python -m
torch.distributed.run`, a child process created before anything here touches the GPU) and exits with
their code; under a launcher (WORLD_SIZE set) it is one of the ranks and WORLD_SIZE must equal --gpus.

A *step* is one `iterate` of the hot path over one batch: n_iter=100 CHOMP iterations (costs every
iteration, final cost pass included).  The workload:

  N = 1 (default)   BASELINE configs[1]: WAM 7-DOF, n_points=100, batch 1024 random adofgoal, fp64
  N > 1             BASELINE configs[2]: the same runs, batch 65 536 drawn with seed 20250102 and cut
                    into contiguous blocks of 8 192 per GPU (rank r iterates block r; weak scaling in
                    N, no data-path collective); the trajectories of step 0 are gathered on the host
                    of rank 0 afterwards and that time is reported separately (`gather`)
  --config 4 / 5    the other two single-GPU configurations as bench lines of their own; the default
                    run (N = 1, config 2) appends both as `other_configs` (5 steps each, with their own
                    `roofline`, counter traffic and `cpu_baseline`) so that they sit under the same clock

Every step works on its own freshly created batch (created before the timed region; the
trajectories are resident in HBM when it starts).  `value` = iterations the runs actually made
(the kernel counts them per run: a run that leaves its joint limits stops for the rest of the call,
as the reference throws) / wall time of the K steps, which are issued round-robin on `--streams`
HIP streams (default 2: two launches overlap, so that the tail of one -- a few slow runs -- is filled by
the next; more streams give the same throughput and longer launches) .
`value_serial` is the same workload with strictly serial launches on one stream.

`roofline`: the contract's yardstick -- ALGORITHMIC bytes of SURVEY.md 8(d) per launch / the fused
kernel's average duration (HIP events on the launch stream) against the 8 TB/s HBM peak -- plus what
the counters say actually binds the kernel (`bound`, `valu_issue`, from profiles/counters_latest.json).
`cpu_baseline` times the oracle (oracle/, a CPU restatement of the reference) on the host cores over
a bounded sample of the same workload.  The process exits non-zero when the parity spot check
against the oracle exceeds its bound.
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

N_ITER = 100
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
CLOCK_HZ = 2.4e9
N_SIMD = 256 * 4
FLOP_PER_ITERATION = {2: 0.8e6, 3: 0.8e6}      # SURVEY.md 8(d) "Algorithmic flops", config W


def algorithmic_bytes_per_iter(m, n, Sa, n_sdf, w, momentum):
    """2 m n w (T) [+ 2 m n w momentum] + m Sa Nsdf 4 w (SDF gathers) + 3 w (costs)."""
    return 2 * m * n * w + (2 * m * n * w if momentum else 0) + m * Sa * n_sdf * 4 * w + 3 * w


def host_cores():
    cores = os.cpu_count() or 1
    try:                                  # the container's CPU quota (cgroup v2), when there is one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return min(cores, len(os.sched_getaffinity(0)))


COMPACT_LIMIT = 4096            # the driver's record parses the LAST stdout line; it must stay small (round-5 review: a 27 KB line did not parse)


def _r(v, digits=5):
    """numbers at a few significant digits: the compact line is for a parser, the full record keeps every bit"""
    if v is None or isinstance(v, (bool, int, str)):
        return v
    try:
        return float("%.*g" % (digits, float(v)))
    except (TypeError, ValueError):
        return None


def compact_line(out, full_path=None):
    """The one JSON object the driver parses: the contract's keys, `roofline` and `cpu_baseline` as numbers, the parity
    figures and `summary`.  Everything else of `out` (other_configs, batch_sweep, tsr*, held4, stages, the prose notes)
    lives in the full record (`--full-out`)."""
    rf = out.get("roofline") or {}
    vi = rf.get("valu_issue") or {}
    cpu = out.get("cpu_baseline")
    cfg = out.get("config") or {}
    line = {
        "metric": out.get("metric"), "value": _r(out.get("value"), 7), "unit": out.get("unit"), "n_gpus": out.get("n_gpus"),
        "steps": out.get("steps"), "warmup": out.get("warmup"), "ms_per_step": _r(out.get("ms_per_step"), 6),
        "higher_is_better": True, "scaling": out.get("scaling"), "vs_baseline": None, "dtype": out.get("dtype"),
        "data": out.get("data"),
        "config": {"workload": str(cfg.get("workload"))[:200], "runs_per_gpu": cfg.get("runs_per_gpu"), "n_iter": cfg.get("n_iter"),
                   "n_points": cfg.get("n_points"), "dof": cfg.get("dof"), "parallelism": cfg.get("parallelism")},
        "value_serial": _r(out.get("value_serial"), 7),
        "iterations_made": out.get("iterations_made"), "iterations_nominal": out.get("iterations_nominal"),
        "runs_outside_joint_limits": out.get("runs_outside_joint_limits"),
        "roofline": {"bound": "hbm", "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
                     "frac": _r(rf.get("frac")), "traffic": _r(rf.get("traffic")),
                     "hbm_measured_frac": _r(rf.get("hbm_measured_frac")), "kernel": str(rf.get("kernel"))[:80],
                     "avg_kernel_ms": _r(rf.get("avg_kernel_ms")), "launches": rf.get("launches"),
                     "concurrent_launches": rf.get("concurrent_launches"),
                     "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
                     "counters_say": rf.get("bound"), "valu_issue": {"frac": _r(vi.get("frac"))} if vi else None,
                     "fp64_vector_frac": _r((rf.get("fp64_vector") or {}).get("frac"))},
        "cpu_baseline": None if not cpu else {"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"),
                                              "value_1_core": _r(cpu.get("value_1_core")), "host_cores": cpu.get("host_cores"),
                                              "kind": cpu.get("kind"), "sample": str(cpu.get("sample"))[:160]},
        "parity_rel_l2_max_vs_oracle": _r(out.get("parity_rel_l2_max_vs_oracle"), 3), "parity_bound": out.get("parity_bound"),
        "parity_runs_checked": out.get("parity_runs_checked"),
        "parity_ill_conditioned_runs": len(out.get("parity_ill_conditioned_runs") or []),
    }
    if out.get("per_rank"):
        line["per_rank_value"] = [_r(p["value"]) for p in out["per_rank"]]
    if out.get("gather"):
        line["gather_s"] = _r(out["gather"].get("total_s"))
    if out.get("backend"):
        line["backend"] = out["backend"]
    if full_path:
        line["full_record"] = os.path.basename(full_path)
    if out.get("summary"):
        line["summary"] = out["summary"]
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:                 # never grow past the limit again: drop the optional parts, widest first
        for k in ("summary", "per_rank_value", "cpu_baseline.sample", "config.workload"):
            if "." in k:
                a, b = k.split(".")
                if line.get(a):
                    line[a][b] = str(line[a][b])[:60]
            elif k == "summary" and isinstance(line.get(k), dict):
                line[k].pop("sweep", None)
            else:
                line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= COMPACT_LIMIT:
                break
    assert len(text) <= COMPACT_LIMIT, len(text)
    return text


class Workload:
    """one BASELINE configuration: how to set the scene up, create a batch, and run the oracle on it"""

    def __init__(self, config, rank, world, batch):
        import common
        from or_cdchomp_amd import robots
        self.config = config
        self.common = common
        self.precision = 64
        if config in (2, 3):
            self.n_runs = batch or (1024 if config == 2 else 8192)
            if config == 2:
                self.goals = None                       # per step: its own seed
            else:
                self.goals = common.config3_goals(rank=rank, world=8) if self.n_runs == 8192 else \
                    common.wam_goals(65536, seed=20250102)[rank * self.n_runs:(rank + 1) * self.n_runs]
            self.kw = dict(common.CONFIG2_KW)
            self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum = 98, 7, 15, 1, 8, False
            self.dtype = "f64"
            self.kernel = "chomp_iterate_kernel<double, chain, 16-lane rows>"
            self.label = ("WAM 7-DOF, n_points=100, batch=%d random adofgoal per GPU, n_iter=%d per step, lambda=100 "
                          "obs_factor=500, tabletop SDF 40x31x12 (BASELINE configs[%d]%s)"
                          % (self.n_runs, N_ITER, 1 if config == 2 else 2,
                             "" if config == 2 else ": block %d of the 65536-run batch, seed 20250102" % rank))
        elif config == 4:
            self.n_runs = batch or 4096
            self.goals, self.basegoals, self.seeds, self.kw = common.config4_problem(self.n_runs)
            self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum = 198, 14, 16, 1, 8, True
            self.dtype = "f64"
            self.kernel = "chomp_iterate_kernel<double, chain with a floating base, 16-lane rows>"
            self.label = ("floating base + WAM arm (n=14), n_points=200, use_momentum use_hmc hmc_resample_lambda=0.02 "
                          "seed=run index, batch=%d, n_iter=%d per step (BASELINE configs[3])" % (self.n_runs, N_ITER))
        elif config == 5:
            self.n_runs = batch or 4096
            self.goals = common.config5_goals(self.n_runs)
            self.kw = dict(common.CONFIG5_KW)
            self.precision = 32
            self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum = 198, 30, 60, 4, 4, False
            self.dtype = "f32"
            self.kernel = "chomp_iterate_kernel<float, tree, generic cost path>"
            self.label = ("30-DOF tree, 60 spheres, 4 box kinbodies with fields at cube_extent=0.005, n_points=200, "
                          "batch=%d, fp32, n_iter=%d per step (BASELINE configs[4])" % (self.n_runs, N_ITER))
        elif config in ("tsr1", "tsr3"):
            # the WAM of config 2 with its end-effector link held on a task space region: the hard-constraint step of
            # cd_chomp_iterate (src/libcd/chomp.c:550-600) with con_tsr (src/orcdchomp_mod.cpp:1330-1497) on every point
            self.n_runs = batch or 1024
            self.goals = None
            self.rows = 1 if config == "tsr1" else 3
            self.kw = dict(n_points=100, lambda_=100.0, obs_factor=200.0)
            self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum = 98, 7, 15, 1, 8, False
            self.dtype = "f64"
            self.kernel = "chomp_iterate_kernel<double, chain, 16-lane rows> with the constraint phase (csrc/tsr.h)"
            self.label = ("WAM 7-DOF, n_points=100, batch=%d goals within 0.4 rad of the start, con_tsr 'all link wam7' with %d constrained "
                          "row(s) per moving point, n_iter=%d per step, lambda=100 obs_factor=200" % (self.n_runs, self.rows, N_ITER))
        elif config in ("d2", "d3"):
            # config 2's runs with `derivative 2` / `3` (src/libcd/chomp.c:239-340: a penta- / hepta-diagonal smoothness metric; the
            # reference multiplies by its dense inverse): the band inverse through its rank-D generators, D prefix and D suffix wave
            # scans per column (DESIGN.md section 3 "metric")
            self.n_runs = batch or 1024
            self.goals = None
            self.kw = dict(common.CONFIG2_KW, derivative=int(config[1]))
            self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum = 98, 7, 15, 1, 8, False
            self.dtype = "f64"
            self.kernel = "chomp_iterate_kernel<double, chain, 16-lane rows>, band metric by wave scans"
            self.label = ("WAM 7-DOF, n_points=100, batch=%d random adofgoal, derivative %s, n_iter=%d per step, lambda=100 obs_factor=500"
                          % (self.n_runs, config[1], N_ITER))
        elif config == "held4":
            # config 2's WAM HOLDING a four-sphere box (RobotBase::Grab; src/orcdchomp_mod.cpp:2168-2300: the held body's spheres
            # join the run's list): 15 + 4 = 19 active spheres, the 32-lane kernel family with the dense pair list (csrc/cost_pairs.h)
            self.n_runs = batch or 1024
            self.goals = None
            self.kw = dict(common.CONFIG2_KW)
            self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum = 98, 7, 19, 1, 8, False
            self.dtype = "f64"
            self.kernel = "chomp_iterate_kernel<double, chain, 32-lane groups, dense self-collision pair list>"
            self.label = ("WAM 7-DOF holding a four-sphere box in its hand (19 active spheres), n_points=100, batch=%d random adofgoal, "
                          "n_iter=%d per step, lambda=100 obs_factor=500, tabletop SDF 40x31x12" % (self.n_runs, N_ITER))
        else:
            raise SystemExit("--config must be 2, 3, 4, 5, tsr1, tsr3, held4, d2 or d3")
        self.bytes_iter = algorithmic_bytes_per_iter(self.m, self.n, self.Sa, self.n_sdf, self.w, self.momentum)
        if config in ("tsr1", "tsr3"):
            # + the constraint step: h and J written and read (2 K (n + 1) w), the rows of C' and r' of the block
            # elimination written and read (2 m n (n + 1) w); K = m rows
            K = self.m * self.rows
            self.bytes_iter += 2 * K * (self.n + 1) * self.w + 2 * self.m * self.n * (self.n + 1) * self.w
        self.robots = robots

    TSR_BASE = [-1.0, 0.0, 1.0, 0.0, float(np.sqrt(0.5)), 0.0, float(np.sqrt(0.5))]      # (a unit quaternion: the constraint frames are poses)

    def setup(self, mod):
        if self.config == 5:
            self.model = self.common.setup_product_tree30(mod)
        elif self.config in ("tsr1", "tsr3"):
            from or_cdchomp_amd import scenes
            model, _, dofvals, adofs = self.common.wam_state()
            mod.add_robot(model, transform=self.TSR_BASE, dof_values=dofvals, active_dofs=adofs)
            scenes.add_tabletop(mod)
            mod.SendCommand("computedistancefield kinbody table")
            self.model = model
            self.tsr = None
        elif self.config == "held4":
            self.model, self.hand, self.held_pose = self.common.setup_product_wam_held4(mod)
        else:
            self.model = self.common.setup_product_wam(mod)

    def tsr_spec(self):
        """the end-effector link's own start frame as the TSR frame: xyz free within +-1 m where Bw says so"""
        if self.tsr is None:
            _, _, dofvals, _ = self.common.wam_state()
            R, t = self.model.link_frames(self.TSR_BASE, dofvals)
            li = self.model.link_names.index("wam7")
            Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0] if self.rows > 1 else [-3, 3], [0, 0] if self.rows > 2 else [-3, 3], [-3, 3]]
            self.tsr = (self.robots.Tsr(T0w_R=R[li], T0w_d=t[li], Bw=Bw), li, R[li], t[li], Bw)
        return self.tsr

    def step_goals(self, step, rank):
        if self.config in (2, "held4", "d2", "d3"):
            return self.common.wam_goals(self.n_runs, seed=20250101 + 1000 * rank + step)
        if self.config in ("tsr1", "tsr3"):
            rng = np.random.default_rng(20250105 + 1000 * rank + step)
            return np.ascontiguousarray(np.array(self.robots.WAM_START)[None, :] + 0.4 * rng.uniform(-1, 1, size=(self.n_runs, 7)))
        return self.goals

    def create(self, mod, step, rank):
        g = self.step_goals(step, rank)
        if self.config == 4:
            return mod.batch_create(self.model.name, g, basegoals=self.basegoals, seeds=self.seeds, **self.kw)
        if self.config == 5:
            return mod.batch_create(self.model.name, g, precision=32, **self.kw)
        if self.config in ("tsr1", "tsr3"):
            self._keep = getattr(self, "_keep", []) + [g]
            return int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda %r obs_factor %r "
                                       "con_tsr 'all link wam7' '%s'" % (self.model.name, self.n_runs, g.ctypes.data, self.kw["n_points"],
                                                                       self.kw["lambda_"], self.kw["obs_factor"], self.tsr_spec()[0].serialize())))
        return mod.batch_create(self.model.name, g, **self.kw)

    def oracle_setup(self, O):
        c = self.common
        if self.config == 5:
            _, grids, poses = c.config5_oracle_fields(O)
            self.o_args = (O.OraRobot(self.model), [0.0] * 6 + [1.0], np.zeros(self.model.n_dof), list(range(self.model.n_dof)))
            self.o_fields = (grids, poses)
        else:
            prob = c.tabletop_problem(O)
            _, base, dofvals, adofs = c.wam_state()
            if self.config in ("tsr1", "tsr3"):
                base = self.TSR_BASE
            grabbed = [(self.hand, self.held_pose, c.HELD4_POS, c.HELD4_RAD)] if self.config == "held4" else []
            self.o_args = (O.OraRobot(self.model, grabbed=grabbed), base, dofvals, adofs)
            self.o_fields = ([prob["sdf"]], [prob["pose"]])

    def oracle_run(self, O, idx, goals, threads, scale=1.0):
        rob, base, dofvals, adofs = self.o_args
        if self.config in ("tsr1", "tsr3"):
            # constrained runs: one oracle run object each (the constraint is added to the run), host threads over runs
            from concurrent.futures import ThreadPoolExecutor
            _, li, R, t, Bw = self.tsr_spec()

            def one(k):
                run = O.OraRun(rob, base, dofvals, adofs, goals[k] * scale, self.o_fields[0], self.o_fields[1], O.default_params(**self.kw))
                run.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], O.pose_from_dR(t, R), [0, 0, 0, 0, 0, 0, 1], Bw)
                st, costs = run.iterate(N_ITER)
                out = (run.traj().copy(), np.asarray(costs, dtype=float), st)
                run.destroy()
                return out
            nt = max(1, int(threads))
            with ThreadPoolExecutor(max_workers=nt) as ex:
                res = list(ex.map(one, [int(k) for k in idx]))
            return (np.array([r[0] for r in res]), np.array([r[1] for r in res]), np.array([r[2] for r in res], dtype=np.int32), min(nt, len(res)))
        kw = {}
        if self.config == 4:
            kw = dict(basegoals=self.basegoals[idx], seeds=self.seeds[idx])
        okw = dict(self.kw)
        if "derivative" in okw:
            okw["D"] = okw.pop("derivative")              # (the oracle's name for it)
        return O.batch_run(rob, base, dofvals, adofs, goals[idx] * scale, self.o_fields[0], self.o_fields[1],
                           O.default_params(**okw), N_ITER, max_threads=threads, **kw)


def free_port():
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    return port


def launch_ranks(args, argv):
    """`bench.py --gpus N` outside a launcher: start the N ranks as a CHILD process (torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1) before this process has touched the GPU, and exit with its code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    sys.exit(subprocess.call(cmd, env=env))


def run_workload(config, args, rank, world, device, dist, steps, warmup, serial_steps, batch=0, want_gather=True, want_cpu=True):
    """one bench line: K steps of one BASELINE configuration on this rank's GPU; returns (line or None, exit code)"""
    import torch
    import or_cdchomp_amd
    from or_cdchomp_amd import sharding

    wl = Workload(config, rank, world, batch)
    mod = or_cdchomp_amd.Module(device)
    wl.setup(mod)
    n_runs = wl.n_runs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(steps, warmup, streams):
        """(elapsed seconds max over ranks, iterations made by this rank, batch ids, per-step results)"""
        mod.set_num_streams(streams if streams > 1 else 0)
        # The overlapping leg sets no knob: the planner chooses from the module's stream count, the robot and the run
        # parameters (four workgroups per CU for fixed-base chains, the 128-thread shape for TSR-constrained runs; batch.cpp).
        # One launch at a time of 769..1024 runs is the one case a caller has to SAY, because it is a property of the batch:
        # all of them resident at once with four 192-thread workgroups per CU at 168 registers (orc_set_workgroup_threads)
        serial_1024 = (streams <= 1 and config in (2, 3) and 768 < n_runs <= 1024)
        threads = (192 if serial_1024 else 0) if args.workgroup_threads < 0 else args.workgroup_threads
        mod.set_workgroup_threads(threads)
        wgs_auto = 4 if (config in (2, 3) and streams <= 1 and n_runs > 1024) else 0      # (serial launches of thousands of runs: also the batch's property)
        wgs = wgs_auto if args.workgroups_per_cu < 0 else args.workgroups_per_cu
        mod.set_workgroups_per_cu(wgs)
        knobs = {"orc_set_num_streams": streams if streams > 1 else 0, "orc_set_workgroup_threads": threads,
                 "orc_set_workgroups_per_cu": wgs}
        warm = [wl.create(mod, 900000 + k, rank) for k in range(warmup)]
        timed = [wl.create(mod, k, rank) for k in range(steps)]
        for bid in warm:
            mod.batch_iterate_async(bid, N_ITER)
        for bid in warm:
            mod.batch_sync(bid)
        for bid in warm:
            mod.batch_destroy(bid)
        mod.kernel_time(reset=True)
        barrier()
        t0 = time.perf_counter()
        for bid in timed:
            mod.batch_iterate_async(bid, N_ITER)
        results = [mod.batch_sync(bid) for bid in timed]      # costs + status come back with iterate
        barrier()
        t1 = time.perf_counter()
        elapsed = sharding.max_over_ranks(t1 - t0, dist)
        iters = [mod.batch_iterations_done(bid) for bid in timed]
        kernel_ms, launches = mod.kernel_time()
        return dict(elapsed=elapsed, own_elapsed=t1 - t0, iters=iters, ids=timed, results=results, kernel_ms=kernel_ms, launches=launches,
                    knobs=knobs)

    # ---- strictly serial launches on one stream (reported beside the headline) --------------------
    serial = None
    if serial_steps > 0 and args.streams > 1:
        serial = timed_leg(serial_steps, min(warmup, 2), 1)
        for bid in serial["ids"]:
            mod.batch_destroy(bid)

    # ---- the headline leg ----------------------------------------------------------------------------
    main_leg = timed_leg(steps, warmup, args.streams)
    timed = main_leg["ids"]
    elapsed = main_leg["elapsed"]

    # iterations actually made, all ranks (host-side gather, gloo; no RCCL data path)
    local = {"status": np.concatenate([st for _, st in main_leg["results"]]),
             "iters": np.concatenate(main_leg["iters"]).astype(np.int64),
             "rank_elapsed": np.asarray([main_leg["own_elapsed"]])}
    if serial is not None:
        local["iters_serial"] = np.concatenate(serial["iters"]).astype(np.int64)
    hg = sharding.host_group(dist)
    whole = sharding.gather_host(local, dist, hg)

    # ---- N > 1: the host-side gather of the trajectories of step 0, timed on its own -----
    gather = None
    if world > 1 and want_gather:
        barrier()
        g0 = time.perf_counter()
        traj_local = mod.batch_gettraj(timed[0])                         # device -> host of this rank
        g1 = time.perf_counter()
        allt = sharding.gather_host({"traj": traj_local}, dist, hg)      # hosts -> rank 0
        g2 = time.perf_counter()
        d2h = sharding.max_over_ranks(g1 - g0, dist)
        tot = sharding.max_over_ranks(g2 - g0, dist)
        if rank == 0:
            gather = {"trajectories_bytes": int(allt["traj"].nbytes), "runs": int(allt["traj"].shape[0]),
                      "device_to_host_s": d2h, "total_s": tot,
                      "note": "trajectories [runs][n_points][n] of step 0: hipMemcpy per rank, then a gloo gather_object on "
                              "rank 0; outside the timed region, reported separately"}
            if args.dump_gather:
                np.save(args.dump_gather, allt["traj"])

    # ---- parity spot check against the oracle on the first timed batch (untimed) ---------------------
    parity = None
    parity_ill = []
    k_check = 0
    cpu = None
    parity_bound = 1e-3 if wl.precision == 32 else 1e-6
    rc = 0
    if rank == 0:
        from oracle import oracle_py as O
        O.build(ref=False)
        wl.oracle_setup(O)
        goals0 = wl.step_goals(0, 0)
        traj0 = mod.batch_gettraj(timed[0])
        st0 = main_leg["results"][0][1]
        k_check = min(16 if config in (2, 3) else 8, n_runs)
        idx = np.unique(np.linspace(0, n_runs - 1, k_check).astype(np.int64))
        k_check = len(idx)
        otraj, ocosts, ost, _ = wl.oracle_run(O, idx, goals0, k_check)
        # Some runs of these workloads are chaotic in the reference algorithm itself (CHOMP bouncing off
        # joint limits amplifies rounding x4 per projection): the oracle run again
        # with the goals moved by ONE ulp (up, down) shows which, and how far such a run may legitimately drift
        # (tests/common.py CHAOS_FACTOR, profiles/r04_chaos_ratio.txt).
        # Parity is quoted over the well-conditioned runs; the others are listed with both figures.
        errs = [wl.common.rel_l2(traj0[k], otraj[j]) for j, k in enumerate(idx)]
        self_amp = [0.0] * k_check
        pst = np.zeros(k_check, dtype=np.int64)
        for f in wl.common.ULPS:
            ptraj, _, ps, _ = wl.oracle_run(O, idx, goals0, k_check, scale=f)
            self_amp = [max(self_amp[j], wl.common.rel_l2(ptraj[j], otraj[j])) for j in range(k_check)]
            pst |= (np.asarray(ps) != 0)
        well = [j for j in range(k_check) if self_amp[j] < 1e-9 and ost[j] == 0 and pst[j] == 0 and st0[idx[j]] == 0]
        parity = max(errs[j] for j in well) if well else None
        parity_ill = [{"run": int(idx[j]), "hip_vs_oracle": errs[j], "oracle_vs_oracle_goal_plus_minus_one_ulp": self_amp[j],
                       "status_hip": int(st0[idx[j]]), "status_oracle": int(ost[j])}
                      for j in range(k_check) if j not in well]
        if parity is None or parity > parity_bound:
            rc = 3
        for j in range(k_check):
            if j not in well and ost[j] == 0 and st0[idx[j]] == 0 and errs[j] > max(parity_bound, wl.common.CHAOS_FACTOR * self_amp[j]):
                rc = 3

        if want_cpu and not args.no_cpu_baseline and world == 1:         # the CPU baseline is reported at N=1 only
            cores = host_cores()
            # ~0.1 s (config 2) to ~2.5 s (config 5) per run of 100 iterations on one core; 10-20 s of wall time
            per_core = {2: 48, 3: 48, 4: 12, 5: 6, "tsr1": 24, "tsr3": 4, "held4": 32, "d2": 48, "d3": 48}[config]
            sample = args.cpu_runs or int(min(n_runs, max(8, per_core * cores)))
            sidx = np.arange(min(sample, n_runs))
            c0 = time.perf_counter()
            _, _, _, threads = wl.oracle_run(O, sidx, goals0, cores)
            c1 = time.perf_counter()
            one = np.arange(min(8 if config in (2, 3) else 2, n_runs))
            s0 = time.perf_counter()
            wl.oracle_run(O, one, goals0, 1)
            s1 = time.perf_counter()
            cpu = {"value": len(sidx) * N_ITER / (c1 - c0), "unit": "CHOMP iterations/s", "cores": int(threads),
                   "value_1_core": len(one) * N_ITER / (s1 - s0), "host_cores": int(cores),
                   "kind": "port",
                   "sample": "%d of the %d runs of step 0 x %d iterations, oracle (C restatement of libcd + "
                             "sphere cost, dense A^-1 as the reference, fp64), OpenMP over runs, %.1f s wall"
                             % (len(sidx), n_runs, N_ITER, c1 - c0)}

    out = None
    if rank == 0:
        made = float(whole["iters"].sum())
        nominal = float(world) * n_runs * N_ITER * steps
        value = made / elapsed
        avg_ms = main_leg["kernel_ms"] / max(main_leg["launches"], 1)
        bytes_iter = wl.bytes_iter
        bytes_launch = bytes_iter * (float(main_leg["iters"][0].sum()) if main_leg["iters"] else 0.0)   # what launch 0 really made
        bytes_launch_nominal = bytes_iter * n_runs * N_ITER
        achieved = bytes_iter * (made / world / steps) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        value_serial = None
        serial_ms = None
        if serial is not None:
            value_serial = float(whole["iters_serial"].sum()) / serial["elapsed"]
            serial_ms = serial["kernel_ms"] / max(serial["launches"], 1)
        per_rank = None
        if world > 1:
            per = whole["iters"].reshape(world, -1).sum(axis=1)
            per_rank = [{"rank": r, "iterations_made": float(per[r]), "elapsed_s": float(whole["rank_elapsed"][r]),
                         "value": float(per[r] / whole["rank_elapsed"][r])} for r in range(world)]
        # counters of profiles/ (rocprofv3 --pmc passes of this command, scripts/pmc_counters.sh)
        traffic = None
        valu = None
        scratch_note = None
        counters_note = "no counters under profiles/ for this workload"
        cpath = os.path.join(ROOT, "profiles", "counters_latest.json")
        if os.path.exists(cpath):
            try:
                from or_cdchomp_amd import _capi
                cj = json.load(open(cpath)).get("config%s" % (2 if config == 3 else config))
                if cj:
                    scratch_note = cj.get("traffic_note")
                if cj and cj.get("csrc_hash") != _capi.csrc_hash():
                    # counters of another build say nothing about this one
                    counters_note = ("profiles/counters_latest.json was taken from build %s, this is build %s: traffic and valu_issue "
                                     "are not quoted (scripts/profile_round.sh + scripts/summarize_profile.py renew them)"
                                     % (cj.get("csrc_hash"), _capi.csrc_hash()))
                    cj = None
                if cj and cj.get("batch") == n_runs and cj.get("n_iter") == N_ITER:
                    counters_note = "rocprofv3 --pmc passes of this build (%s), %s" % (cj.get("csrc_hash"), cj.get("source"))
                    traffic = cj.get("hbm_bytes_per_launch")
                    ipri = cj.get("valu_insts_per_run_iteration")
                    if ipri:
                        # a wave64 vector instruction holds its SIMD for 4 cycles (16 lanes per cycle)
                        busy = ipri * 4.0 * (value / world) / (N_SIMD * CLOCK_HZ)
                        valu = {"valu_wave_insts_per_run_iteration": ipri, "issue_cycles_per_inst": 4,
                                "frac": busy, "simds": N_SIMD, "clock_hz": CLOCK_HZ,
                                "source": cj.get("source"),
                                "note": "SQ_INSTS_VALU per launch / (runs x iterations) x 4 cycles x iterations/s "
                                        "/ (1024 SIMDs x clock): the share of vector issue slots the kernel fills"}
            except Exception:
                traffic, valu, scratch_note = None, None, None
        if traffic is None:
            scratch_note = None
        flop = FLOP_PER_ITERATION.get(config)
        out = {
            "metric": "CHOMP iters/sec, 7-DOF x 100-waypoint" if config in (2, 3) else (
                "CHOMP iters/sec, 7-DOF x 100-waypoint, the robot holds a four-sphere body" if config == "held4" else
                "CHOMP iters/sec, 7-DOF x 100-waypoint, derivative %s" % config[1] if config in ("d2", "d3") else
                "CHOMP iters/sec, 7-DOF x 100-waypoint, TSR-constrained (%s)" % config if isinstance(config, str)
                else "CHOMP iters/sec (BASELINE configs[%d])" % (config - 1)),
            "value": value,
            "unit": "CHOMP iterations/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic",
            "config": {"workload": wl.label, "runs_per_gpu": n_runs, "n_iter": N_ITER, "n_points": wl.m + 2, "dof": wl.n,
                       "parallelism": "runs sharded over %d GPU(s) in contiguous blocks, no collective" % world,
                       # the module settings of each leg (include/orcdchomp_amd.h); a caller with the defaults (all 0) gets the
                       # planner's own shape: trajectories are bit-identical, throughput a few per cent lower
                       "knobs": {"value": main_leg["knobs"], "value_serial": None if serial is None else serial["knobs"]}},
            "backend": None if dist is None else dist.get_backend(),
            "dist_world_size": 1 if dist is None else dist.get_world_size(),
            "iterations_made": made, "iterations_nominal": nominal,
            "runs_outside_joint_limits": int((whole["status"] != 0).sum()),
            "runs_total": int(whole["status"].size),
            "per_rank": per_rank,
            "value_serial": value_serial,
            "value_serial_note": None if serial is None else
                "%d steps, strictly serial launches on one stream, avg kernel %.2f ms%s" % (serial_steps, serial_ms,
                    " (192-thread workgroups, four per CU: the %d runs of a launch are resident at once)" % n_runs
                    if (config in (2, 3) and 768 < n_runs <= 1024) else ""),
            "roofline": {"bound": "valu" if valu else "hbm",
                         "yardstick": "hbm: SURVEY.md 8(d) algorithmic bytes per launch / average launch duration vs the 8 TB/s peak "
                                      "(the contract's figure); `bound` names what the counters say binds the kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # what `traffic` is made of: the run's state lives in LDS and the field in L2, so nearly all of it is the
                         # callee-saved registers of the phase calls going to scratch and back (writes - results), not state
                         "traffic_note": scratch_note,
                         # what the memory system really moved: counter bytes per launch / launch duration vs the peak
                         # (well above `frac`: the kernel re-reads or spills; below: the state stays on chip)
                         "hbm_measured_frac": None if not (traffic and avg_ms > 0) else traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "kernel": wl.kernel, "avg_kernel_ms": avg_ms, "launches": main_leg["launches"],
                         "concurrent_launches": max(1, args.streams),
                         "serial_avg_kernel_ms": serial_ms,
                         "serial_frac": None if not serial_ms else
                             bytes_iter * (float(whole["iters_serial"].sum()) / world / serial_steps) / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         # all overlapping launches together: algorithmic bytes of the K steps / their wall time
                         "aggregate_achieved": value / world * bytes_iter / 1e9,
                         "aggregate_frac": value / world * bytes_iter / 1e9 / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": bytes_launch_nominal,
                         "algorithmic_bytes_launch_0_as_made": bytes_launch,
                         "algorithmic_bytes_per_iteration_per_run": bytes_iter,
                         "valu_issue": valu, "counters": counters_note},
            "cpu_baseline": cpu,
            "parity_rel_l2_max_vs_oracle": parity, "parity_bound": parity_bound,
            "parity_runs_checked": k_check, "parity_ill_conditioned_runs": parity_ill,
            "gather": gather,
        }
        if flop:
            out["roofline"]["fp64_vector"] = {"achieved": value * flop / 1e12 / world, "peak": FP64_VECTOR_PEAK_TFLOPS,
                                              "unit": "TFLOP/s", "frac": value * flop / 1e12 / world / FP64_VECTOR_PEAK_TFLOPS,
                                              "algorithmic_flop_per_iteration_per_run": flop}
    for bid in timed:
        mod.batch_destroy(bid)
    mod.close()
    return out, rc


def batch_sweep(device, batches):
    """BASELINE.json quotes the metric "batch swept": config 2's runs at other batch sizes, one launch at a time on one
    stream (a launch ends with its slowest run).  Batches of at most 256 runs also with the latency shape
    (orc_set_workgroup_threads(512): eight wavefronts on a run, one run per CU), which is what the single-run `create`
    command uses by itself."""
    import torch
    import common
    import or_cdchomp_amd
    mod = or_cdchomp_amd.Module(device)
    model = common.setup_product_wam(mod)
    lines = []
    for b in batches:
        launches = 10 if b <= 64 else (3 if b <= 4096 else (2 if b <= 16384 else 1))
        entry = {"batch": int(b), "launches": launches, "n_iter": N_ITER}
        for name, threads in (("value", 0), ("value_latency_shape", 512)):
            if threads and b > 256:
                continue
            wgs = 4 if b > 1024 else 0
            mod.set_workgroup_threads(threads)
            mod.set_workgroups_per_cu(wgs)
            warm = mod.batch_create(model.name, common.wam_goals(b, seed=20250301 + b), **common.CONFIG2_KW)
            mod.batch_iterate(warm, N_ITER)
            mod.batch_destroy(warm)
            ids = [mod.batch_create(model.name, common.wam_goals(b, seed=20250401 + 17 * b + k), **common.CONFIG2_KW) for k in range(launches)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for bid in ids:
                mod.batch_iterate_async(bid, N_ITER)
            for bid in ids:
                mod.batch_sync(bid)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            made = float(sum(mod.batch_iterations_done(bid).sum() for bid in ids))
            for bid in ids:
                mod.batch_destroy(bid)
            entry[name] = made / (t1 - t0)
            if not threads:
                entry["ms_per_launch"] = (t1 - t0) / launches * 1e3
                entry["iterations_made"] = made
                entry["knobs"] = {"orc_set_num_streams": 0, "orc_set_workgroup_threads": 0, "orc_set_workgroups_per_cu": wgs}
        lines.append(entry)
    mod.close()
    return {"workload": "config 2's runs (WAM 7-DOF, n_points=100, n_iter=100 per launch, costs every iteration) at other batch sizes, "
                        "serial launches on one stream, wall clock around the launches incl. the copy-back of costs and status",
            "unit": "CHOMP iterations/s", "sweep": lines}


def stage_breakdown(device):
    """The reference's own time breakdown (compile-time DEBUG_TIMING, src/orcdchomp_mod.cpp:2835-2847, src/libcd/chomp.c:446-456,
    662-676) for config 2's batch, from the kernel's per-phase cycle counters (ORC_PHASE_TIMERS=1: s_memtime around the
    phases of every iteration, one workgroup = one run).  The timers cost a few per cent, so this is a step of its own."""
    import ctypes as C
    import common
    import or_cdchomp_amd
    os.environ["ORC_PHASE_TIMERS"] = "1"
    try:
        mod = or_cdchomp_amd.Module(device)
        model = common.setup_product_wam(mod)
        n_runs = 1024
        bid = mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101), **common.CONFIG2_KW)
        mod.batch_iterate(bid, N_ITER)
        ph = np.zeros((n_runs, 8))
        mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", ph.ctypes.data_as(C.POINTER(C.c_double)), ph.size))
        it = mod.batch_iterations_done(bid).astype(np.float64) + 1.0          # + the final cost-only pass
        mod.batch_destroy(bid)
        mod.close()
    finally:
        os.environ.pop("ORC_PHASE_TIMERS", None)
    cyc = ph[:, :6].sum(axis=0) / it.sum()                                    # cycles per run-iteration: FK, cost, obs-reduce, smooth+solve+step, limits, smooth cost
    total = float(cyc.sum())
    sec = lambda c: float(c) / CLOCK_HZ * N_ITER                              # seconds of 100 iterations of one run's workgroup
    return {
        "unit": "seconds one run's workgroup spends per 100 iterations (cycles / 2.4 GHz), mean over the 1024 runs of config 2; "
                "three to four workgroups share a CU, so the shares, not the sums, compare with wall time",
        "cycles_per_iteration": total,
        "ticks_vels": 0.0,
        "ticks_callback_pre": sec(cyc[0]),
        "ticks_fk": sec(cyc[0]), "ticks_jacobians": 0.0, "ticks_pre_velsaccs": 0.0,
        "ticks_callbacks": sec(cyc[1] + cyc[2]),
        "ticks_selfcol": None,
        "ticks_smoothgrad": sec(cyc[3]),
        "ticks_joint_limits": sec(cyc[4]),
        "ticks_smoothcost": sec(cyc[5]),
        "share": {"ticks_callback_pre": float(cyc[0]) / total, "ticks_callbacks": float(cyc[1] + cyc[2]) / total,
                  "ticks_smoothgrad": float(cyc[3]) / total, "ticks_joint_limits": float(cyc[4]) / total,
                  "ticks_smoothcost": float(cyc[5]) / total},
        "mapping": "ticks_vels: the dense velocity product (chomp.c:449-451) is dead work on this path and not made (SURVEY.md 8a M1). "
                   "ticks_callback_pre = ticks_fk: the FK phase (sphere centres, joint axes and anchors of every waypoint); "
                   "ticks_jacobians 0: the 3 x n sphere Jacobians are never formed (J^T is a wrench sum inside the cost pass); "
                   "ticks_pre_velsaccs 0: the central differences are taken inside the cost pass from the positions in LDS. "
                   "ticks_callbacks: the cost pass (field lookups, obstacle and self-collision forces, J^T) and its reduction; "
                   "ticks_selfcol is inside it and has no timer of its own in the product build (-DORC_COST_TIMERS builds have: "
                   "DESIGN.md section 3). ticks_smoothgrad: G += A T + B TOGETHER WITH what the reference leaves untimed: A^-1 G, "
                   "the step (chomp.c:525-605); ticks_joint_limits: the projection rounds (chomp.c:608-655, untimed in the reference); "
                   "ticks_smoothcost: chomp.c:660-677",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="", help="BASELINE configuration: 2 (default at --gpus 1), 3 (default at --gpus > 1), 4 or 5; d2 / d3: config 2 with derivative 2 / 3; "
                                                 "tsr1 / tsr3: config 2's runs held on a TSR by one / three hard-constraint rows "
                                                 "on every moving point (con_tsr, SURVEY.md 8f rank 4); held4: config 2's WAM holding a four-sphere box")
    ap.add_argument("--batch", type=int, default=0, help="runs per GPU (default: the configuration's own size)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin: consecutive steps are independent batches, "
                         "so the tail of one launch (a few slow runs) is filled by the next; 1 = strictly serial launches")
    ap.add_argument("--serial-steps", type=int, default=-1, help="steps of the strictly serial leg (value_serial); default min(steps, 10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=0, help="override the cpu baseline sample size")
    ap.add_argument("--backend", default="gloo", help="torch.distributed backend for N>1.  The path has no data-path collective "
                                                      "(north_star: host-side gather only): a barrier, one max-reduce of the elapsed "
                                                      "time and the gather of results, all on the host (gloo).  nccl (= RCCL) is optional")
    ap.add_argument("--workgroups-per-cu", type=int, default=-1, help="register budget of the batches (orc_set_workgroups_per_cu): 0 (the planner's choice), 3 or 4; "
                                                                      "default: 0, and 4 for serial launches of more than 1024 runs of configs 2 and 3")
    ap.add_argument("--workgroup-threads", type=int, default=-1, help="workgroup shape of the batches (orc_set_workgroup_threads): 0, 128, 192, 256 or 512; "
                                                                      "default: the leg's own choice")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run only (N = 1, config 2): do not append the config 4 and config 5 lines")
    ap.add_argument("--other-steps", type=int, default=16, help="steps of each `other_configs` line (even: the steps alternate between two streams; "
                                                               "the first and the last step of a stream overlap with nothing: 16 steps are within 2 %% of 40)")
    ap.add_argument("--no-sweep", action="store_true", help="default run only: do not append `batch_sweep`, `tsr1`, `tsr3` and `stages`")
    ap.add_argument("--sweep-batches", default="1,64,4096,8192,16384,65536", help="batch sizes of `batch_sweep` (8192 = one block of config 3: "
                                                                                  "the N = 1 point of the multi-GPU curve, like for like)")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "bench_full.json"),
                    help="rank 0 writes the FULL record here (every line, sweep, stages, notes); stdout's last line is the compact one")
    ap.add_argument("--print-full", action="store_true", help="also print the full record as an EARLIER stdout line")
    ap.add_argument("--dump-gather", default="", help="N > 1: rank 0 saves the gathered step-0 trajectories here (.npy)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # `--gpus N` outside a launcher starts its own ranks; nothing above or in launch_ranks touches the GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    config = args.config or (2 if world == 1 else 3)
    if str(config) in ("2", "3", "4", "5"):
        config = int(config)

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

    serial_steps = args.serial_steps if args.serial_steps >= 0 else min(args.steps, 10)
    out, rc = run_workload(config, args, rank, world, device, dist, args.steps, args.warmup, serial_steps, batch=args.batch)
    if world == 1 and not args.config and not args.batch and not args.no_other_configs:
        # the other two single-GPU configurations under the same clock (a few steps each)
        others = []
        for c in (4, 5):
            line, rc_c = run_workload(c, args, rank, world, device, dist, args.other_steps, 2, min(args.other_steps, 3))
            others.append(line)
            rc = rc or rc_c
        out["other_configs"] = others
        # the rest of the metric under the same clock: the batch sweep, the TSR-constrained line, the reference's stage names
        if not args.no_sweep:
            out["batch_sweep"] = batch_sweep(device, [int(b) for b in args.sweep_batches.split(",")])
            for c in ("tsr1", "tsr3", "held4", "d2", "d3"):
                line, rc_c = run_workload(c, args, rank, world, device, dist, args.other_steps, 2, min(args.other_steps, 3),
                                          want_cpu=(c not in ("d2", "d3")))
                out[c] = line
                rc = rc or rc_c
            out["stages"] = stage_breakdown(device)
            # one 8192-run block of config 3 on this GPU, issued like the N > 1 line issues it: the N = 1 point of the
            # scaling curve, like for like (the N = 1 headline is config 2's 1024 runs, a different per-GPU workload)
            line, rc_c = run_workload(3, args, rank, world, device, dist, 6, 1, 3, want_gather=False, want_cpu=False)
            out["c3_block"] = line
            rc = rc or rc_c
    if rank == 0 and world == 1 and out is not None:
        # every headline number once more, compact, as the LAST key of the line (a record that keeps only the tail of the line keeps this)
        from or_cdchomp_amd import _capi
        r3 = lambda v: None if v is None else float("%.4g" % v)
        summ = {"unit": "M it/s", "build": _capi.csrc_hash()}
        key = {2: "c2", 3: "c3", 4: "c4", 5: "c5"}.get(config, str(config))
        lines = [(key, out)] + [({4: "c4", 5: "c5"}[4 + i], l) for i, l in enumerate(out.get("other_configs") or [])]
        lines += [(c, out[c]) for c in ("tsr1", "tsr3", "held4", "d2", "d3", "c3_block") if out.get(c)]
        for k, l in lines:
            summ[k] = r3(l["value"] / 1e6)
            summ[k + "_serial"] = r3(None if l.get("value_serial") is None else l["value_serial"] / 1e6)
            summ[k + "_frac"] = r3(l["roofline"]["frac"])
            summ[k + "_parity"] = r3(l.get("parity_rel_l2_max_vs_oracle"))
            if l.get("cpu_baseline"):
                summ[k + "_cpu"] = r3(l["cpu_baseline"]["value"] / 1e6)
        if out.get("batch_sweep"):
            summ["sweep"] = {str(e["batch"]): r3(e["value"] / 1e6) for e in out["batch_sweep"]["sweep"]}
        out["summary"] = summ
    if rank == 0:
        full_path = args.full_out or None
        if full_path:
            try:
                with open(full_path, "w") as f:
                    json.dump(out, f)
                mirror = os.path.join(ROOT, "gpurun_out")           # (a gpurun box brings this directory back)
                if os.path.isdir(mirror):
                    with open(os.path.join(mirror, os.path.basename(full_path)), "w") as f:
                        json.dump(out, f)
            except OSError as e:
                print("bench.py: could not write %s: %s" % (full_path, e), file=sys.stderr)
                full_path = None
        if args.print_full:
            print(json.dumps(out))
        print(compact_line(out, full_path))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
