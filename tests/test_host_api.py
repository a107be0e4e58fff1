"""not gpu: the C ABI loads and exports everything include/orcdchomp_amd.h declares; the command
string builders emit the reference's grammar; the product refuses to run without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    from or_cdchomp_amd import _capi
    header = open(os.path.join(ROOT, "include", "orcdchomp_amd.h")).read()
    declared = set(re.findall(r"\b(orc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    L = _capi.lib()
    bound = {s[0] for s in _capi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(L, name) is not None


def test_no_cpu_fallback():
    """without a HIP device the module refuses loudly instead of computing on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import or_cdchomp_amd
    with pytest.raises(RuntimeError, match="no HIP device available"):
        or_cdchomp_amd.Module(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "or_cdchomp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"(import|from)\s+oracle|oracle/|liboracle|ora_[a-z]+_", text), f


class _Recorder:
    def __init__(self):
        self.cmds = []

    def SendCommand(self, cmd, releasegil=False):
        self.cmds.append(cmd)
        return "7" if cmd.startswith("create") else ("12.5" if cmd.startswith("iterate") else "")


def test_bindings_emit_reference_grammar():
    """key order and number formats of /root/reference pythonsrc/orcdchomp/orcdchomp.py:105-167"""
    from or_cdchomp_amd import bindings
    m = bindings.bind(_Recorder())
    m.computedistancefield(kinbody="table", cube_extent=0.02, aabb_padding=0.2, cache_filename="a b.dat", require_cache=True)
    assert m.cmds[-1] == "computedistancefield kinbody 'table' cube_extent 0.020000 aabb_padding 0.200000 cache_filename 'a b.dat' require_cache"
    m.addfield_fromobsarray(kinbody="k", obsarray="0x1234", sizes=[2, 3, 4], lengths=[0.1, 0.2, 0.3], pose=[0, 0, 0, 0, 0, 0, 1])
    assert m.cmds[-1] == "addfield_fromobsarray kinbody 'k' obsarray 0x1234 sizes '2 3 4' lengths '0.1 0.2 0.3' pose '0 0 0 0 0 0 1'"
    run = m.create(robot="it's", adofgoal=[0.6, -1.2], basegoal=[0, 0, 0, 0, 0, 0, 1], floating_base=True, lambda_=100.0,
                   n_points=100, derivative=1, use_momentum=True, use_hmc=True, hmc_resample_lambda=0.02, seed=3,
                   epsilon=0.1, epsilon_self=0.04, obs_factor=500.0, obs_factor_self=10.0, no_report_cost=True)
    assert run == "7"
    assert m.cmds[-1] == ("create robot 'it'\\''s' adofgoal '0.6 -1.2' basegoal '0 0 0 0 0 0 1' floating_base lambda 100.0000 "
                          "n_points 100 derivative 1 use_momentum use_hmc hmc_resample_lambda 0.020000 seed 3 "
                          "epsilon 0.100000 epsilon_self 0.040000 obs_factor 500.000000 obs_factor_self 10.000000 no_report_cost")
    cost = [None]
    m.iterate(run=run, n_iter=100, max_time=2.5, trajs_fileformstr="t_%03d.xml", cost=cost)
    assert m.cmds[-1] == "iterate run 7 n_iter 100 max_time 2.500000 trajs_fileformstr 't_%03d.xml'" and cost[0] == 12.5
    m.gettraj(run=run, no_collision_check=True, no_collision_exception=True, no_collision_details=True)
    assert m.cmds[-1] == "gettraj run 7 no_collision_check no_collision_exception no_collision_details"
    m.destroy(run=run)
    assert m.cmds[-1] == "destroy run 7"
    m.cmds.clear()
    m.runchomp(robot="r", n_iter=5, lambda_=100.0, obs_factor=500.0, adofgoal=[1, 2], no_collision_exception=True)
    assert [c.split()[0] for c in m.cmds] == ["create", "iterate", "gettraj", "destroy"]
    assert m.cmds[0] == "create robot 'r' adofgoal '1 2' lambda 100.0000 obs_factor 500.000000"


def test_robot_models():
    from or_cdchomp_amd import robots
    w = robots.wam7()
    a = w.arrays()
    assert a["n_spheres"] == 16 and a["n_dof"] == 11
    assert (a["parent"] < range(a["n_links"])).all()
    t = robots.tree30()
    assert t.n_dof == 30 and len(t.spheres) == 60


def _bench(extra, env_extra=None, timeout=300):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_refuses_a_world_that_is_not_gpus():
    """under a launcher (WORLD_SIZE set) --gpus must name that world, in either direction"""
    r = _bench(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE (1) != --gpus (2)" in (r.stdout + r.stderr)
    r = _bench(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE (2) != --gpus (1)" in (r.stdout + r.stderr)


def test_bench_gpus_n_starts_its_own_ranks():
    """`bench.py --gpus 2` with no WORLD_SIZE starts two child ranks itself and exits with their code: without a
    GPU every rank refuses ("needs an MI355X"), so the command must fail -- through the launcher it started
    (the launcher ends the other rank as soon as one has failed: at least one of them has said so)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_multirank.py runs the real thing")
    r = _bench(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    text = r.stdout + r.stderr
    assert text.count("bench.py needs an MI355X") >= 1 and "torch.distributed" in text, text[-3000:]
