"""-m gpu: random command text against the parser of the SendCommand surface.  The reference answers a
malformed command with an exception ("Bad arguments!", src/orcdchomp_mod.cpp:2081-2085, and the messages of 8b); so
must this build: a reply or an error for every string, and a module that still works afterwards."""
import os

import numpy as np
import pytest

import common
from or_cdchomp_amd import scenes

pytestmark = pytest.mark.gpu

VERBS = ["computedistancefield", "addfield_fromobsarray", "removefield", "create", "iterate", "gettraj", "destroy",
         "createbatch", "iteratebatch", "gettrajbatch", "destroybatch", "nonsense", ""]
KEYS = ["kinbody", "robot", "adofgoal", "basegoal", "floating_base", "lambda", "n_points", "derivative", "use_momentum", "use_hmc",
        "hmc_resample_lambda", "seed", "epsilon", "epsilon_self", "obs_factor", "obs_factor_self", "dat_filename", "starttraj",
        "run", "n_iter", "max_time", "trajs_fileformstr", "no_collision_check", "no_collision_exception", "no_collision_details",
        "aabb_padding", "cube_extent", "cache_filename", "require_cache", "sizes", "lengths", "pose", "con_tsr", "start_tsr",
        "everyn_tsr", "n_runs", "precision", "devices", "no_report_cost", "start_cost", "ee_force"]
VALUES = ["table", "mug", "BarrettWAM", "ghost", "0", "1", "-1", "3", "100", "1e9", "20000", "-5", "0.0", "0.001", "nan", "inf", "abc", "''",
          "'0.1 0.2 0.3 0.4 0.5 0.6 0.7'", "'0.1 0.2'", "'a b c'", "'1 2 3'", "'0 0 0 0 0 0 1'", "'", "\"unterminated", "\\",
          "'/nonexistent/dir/file_%d.txt'", "'%s%s%s%n'", "'<trajectory></trajectory>'", "'<trajectory><data count=\"2\">1 2 3</data></trajectory>'",
          "'all'", "'all link wam7'", "'all link ghost'", "'0 NULL 1 0 0 0 1 0 0 0 1 0 0 0'", "99999999999999999999", "-0", "2", "7", "64"]


GOAL = "'0.6 -1.2 0.3 1.6 -0.4 0.5 0.2'"
TEMPLATES = [
    ["computedistancefield", "kinbody", "mug", "cube_extent", "0.02", "aabb_padding", "0.2"],
    ["removefield", "kinbody", "mug"],
    ["create", "robot", "BarrettWAM", "adofgoal", GOAL, "lambda", "100.0", "n_points", "20", "use_momentum", "use_hmc",
     "hmc_resample_lambda", "0.05", "seed", "3", "epsilon", "0.1", "epsilon_self", "0.04", "obs_factor", "200", "obs_factor_self", "10"],
    ["create", "robot", "BarrettWAM", "adofgoal", GOAL, "basegoal", "'-1 0 1 0 0.70711 0 0.70711'", "floating_base", "n_points", "12", "derivative", "2"],
    ["create", "robot", "BarrettWAM", "adofgoal", GOAL, "n_points", "16", "con_tsr", "'all link wam7'", "'0 NULL 1 0 0 0 1 0 0 0 1 0.3 0.2 0.9 1 0 0 0 1 0 0 0 1 0 0 0 -1 1 -1 1 0 0 -3 3 -3 3 -3 3'"],
    ["iterate", "run", "RUN", "n_iter", "3", "max_time", "10.0"],
    ["gettraj", "run", "RUN", "no_collision_check"],
    ["gettraj", "run", "RUN", "no_collision_exception", "no_collision_details"],
    ["destroy", "run", "RUN"],
]


def test_random_command_text_never_faults(tmp_path):
    import or_cdchomp_amd
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    rng = np.random.default_rng(int(os.environ.get("ORC_COMMAND_FUZZ_SEED", "4242")))
    runs = []
    log = open(os.environ["ORC_COMMAND_FUZZ_LOG"], "w") if os.environ.get("ORC_COMMAND_FUZZ_LOG") else None      # the command in flight
    replies = errors = 0
    kinds = {}
    for it in range(int(os.environ.get("ORC_COMMAND_FUZZ", "4000"))):
        verb = VERBS[int(rng.integers(0, len(VERBS)))]
        toks = [verb]
        for _ in range(int(rng.integers(0, 7))):
            r = rng.uniform()
            if r < 0.55:
                toks.append(KEYS[int(rng.integers(0, len(KEYS)))])
            toks.append(VALUES[int(rng.integers(0, len(VALUES)))] if r > 0.25 else "")
        # now and then something that can succeed, so that live handles take part
        if rng.uniform() < 0.1:
            toks = ["create", "robot", model.name, "adofgoal", "'0.6 -1.2 0.3 1.6 -0.4 0.5 0.2'", "n_points", str(int(rng.integers(3, 30)))]
        if runs and rng.uniform() < 0.3:
            h = runs[int(rng.integers(0, len(runs)))]
            toks = [("iterate", "gettraj", "destroy")[int(rng.integers(0, 3))], "run", h] + toks[1:3]
        # and mutations of commands that are right: one token dropped, replaced, doubled, two swapped, the tail cut off
        if rng.uniform() < 0.5:
            h = runs[int(rng.integers(0, len(runs)))] if runs else "1"
            toks = list(TEMPLATES[int(rng.integers(0, len(TEMPLATES)))])
            toks = [h if t == "RUN" else t for t in toks]
            for _ in range(int(rng.integers(0, 3))):
                k = int(rng.integers(1, len(toks)))
                kind = int(rng.integers(0, 5))
                if kind == 0:
                    del toks[k]
                elif kind == 1:
                    toks[k] = VALUES[int(rng.integers(0, len(VALUES)))]
                elif kind == 2:
                    toks.insert(k, toks[k])
                elif kind == 3:
                    j = int(rng.integers(1, len(toks))); toks[k], toks[j] = toks[j], toks[k]
                else:
                    toks = toks[:k]
                if len(toks) < 2:
                    break
        cmd = " ".join(toks)
        if log:
            log.seek(0); log.truncate(); log.write("%d: %s\n" % (it, cmd)); log.flush()
        try:
            out = mod.SendCommand(cmd)
            replies += 1
            if toks[0] == "create" and out.strip().isdigit():
                runs.append(out.strip())
                while len(runs) > 48:                      # (live runs hold device memory: a bounded number of them)
                    mod.SendCommand("destroy run %s" % runs.pop(0))
            if toks[0] == "destroy" and toks[2] in runs:
                runs.remove(toks[2])
        except RuntimeError as e:
            errors += 1
            key = str(e)[:40]
            kinds[key] = kinds.get(key, 0) + 1
            assert str(e) != "", cmd
    assert replies > 400 and errors > 1000, (replies, errors)
    assert kinds.get("Bad arguments!", 0) > 100
    # the module is still what it was: with the scene's fields as they were at the start, the demo run gives the demo's result
    for cmd in ("removefield kinbody mug", "computedistancefield kinbody table"):
        try:
            mod.SendCommand(cmd)
        except RuntimeError:
            pass
    goal = [0.6, -1.2, 0.3, 1.6, -0.4, 0.5, 0.2]
    bid = mod.batch_create(model.name, np.array([goal]), n_points=30, lambda_=100.0, obs_factor=500.0)
    costs, status = mod.batch_iterate(bid, 10)
    fresh = or_cdchomp_amd.Module(0)
    common.setup_product_wam(fresh)
    bid2 = fresh.batch_create(model.name, np.array([goal]), n_points=30, lambda_=100.0, obs_factor=500.0)
    costs2, status2 = fresh.batch_iterate(bid2, 10)
    assert np.array_equal(costs, costs2) and np.array_equal(mod.batch_gettraj(bid), fresh.batch_gettraj(bid2))
    print("random commands: %d replies, %d errors; the most frequent: %s" % (
        replies, errors, sorted(kinds.items(), key=lambda kv: -kv[1])[:6]))
