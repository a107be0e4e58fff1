"""-m gpu: the N > 1 product path (BASELINE configs[2]) rehearsed on ONE card.

`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start its own two ranks (a child
`torch.distributed.run`, created before the parent touches the GPU); each rank iterates its contiguous
block of the 65 536-run batch of seed 20250102 (cut to --batch 512 runs per rank here), rank 0 gathers
the step-0 trajectories on the host (gloo; no data-path collective) and prints the one JSON line.  The
gathered trajectories must equal, bit for bit, a single-process batch of the same 1024 goals: a run's
bits do not depend on which rank or which block it is in.  The log of the rehearsal is kept under
profiles/ (r06_rehearsal_2ranks_gloo.json) when the test runs on the builder's GPU box.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, tmp_path, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_two_ranks_and_gathers(tmp_path):
    dump = str(tmp_path / "gathered.npy")
    full_path = str(tmp_path / "full.json")
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--batch", "512", "--steps", "2", "--warmup", "1",
                    "--no-cpu-baseline", "--dump-gather", dump, "--full-out", full_path], tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints the ONE line
    assert len(lines[0]) <= 4096                                  # ... the compact one (the driver's record parses it)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert len(line["per_rank_value"]) == 2 and all(v > 0 for v in line["per_rank_value"])
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None      # (the CPU baseline is an N = 1 figure)
    assert line["parity_rel_l2_max_vs_oracle"] is not None and line["parity_rel_l2_max_vs_oracle"] <= 1e-6
    full = json.load(open(full_path))                             # the full record beside it
    assert abs(full["value"] - line["value"]) <= 1e-6 * full["value"]
    assert full["gather"]["runs"] == 1024 and full["runs_total"] == 2 * 2 * 512
    assert len(full["per_rank"]) == 2 and all(p["value"] > 0 for p in full["per_rank"])
    got = np.load(dump)
    assert got.shape == (1024, 100, 7)

    # the same 1024 runs as ONE batch in this process
    import or_cdchomp_amd
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    goals = common.wam_goals(65536, seed=20250102)[:1024]
    bid = mod.batch_create(model.name, goals, **common.CONFIG2_KW)
    mod.batch_iterate(bid, 100)
    single = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.close()
    assert np.array_equal(got, single), "gathered trajectories of the two ranks differ from the single-process batch"

    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "r06_rehearsal_2ranks_gloo.json"), "w") as f:
            json.dump({"command": "python bench.py --gpus 2 --backend gloo --batch 512 --steps 2 --warmup 1 --no-cpu-baseline",
                       "line": line, "gathered_equals_single_process_batch": True}, f, indent=1)


def test_in_process_shards_on_distinct_devices():
    """orc_module_new_multi over DISTINCT ordinals (one shard, one stream and one host thread per physical GPU): skipped
    on a box with one card, so that the first 8-GPU box the suite meets exercises the per-device replicas and the
    hipSetDevice-per-thread paths that ordinal 0 repeated cannot.  Bars: the sharded batch equals the single-device batch bit for
    bit (trajectories, costs, status), for the 16-lane family, the pair-list family (a held body) and an hmc run."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs at least two GPUs (this box has %d)" % n_dev)
    import numpy as np
    import common
    import or_cdchomp_amd
    devs = list(range(min(n_dev, 8)))
    n_runs = 64 * len(devs) + 5                       # an uneven cut
    cases = [("wam", dict(common.CONFIG2_KW), {}), ("held4", dict(common.CONFIG2_KW), {}),
             ("hmc", dict(common.CONFIG2_KW, use_momentum=1, use_hmc=1, hmc_resample_lambda=0.05), dict(seeds=np.arange(n_runs, dtype=np.uint32)))]
    for name, kw, extra in cases:
        out = []
        for d in (0, devs):
            mod = or_cdchomp_amd.Module(d)
            model = common.setup_product_wam_held4(mod)[0] if name == "held4" else common.setup_product_wam(mod)
            bid = mod.batch_create(model.name, common.wam_goals(n_runs, seed=91), **kw, **extra)
            costs, status = mod.batch_iterate(bid, 40)
            out.append((mod.batch_gettraj(bid), costs, status))
            mod.batch_destroy(bid)
            mod.close()
        assert np.array_equal(out[0][2], out[1][2]), name
        assert np.array_equal(out[0][0], out[1][0]), name
        assert np.array_equal(out[0][1], out[1][1]), name
