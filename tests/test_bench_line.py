"""The driver parses the LAST stdout line of bench.py: it must be one JSON object of at most 4096 bytes that carries the
contract's keys, `roofline` and `cpu_baseline` (round-5 review: a 27 KB line left `BENCH_r05.json.parsed` null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _check(text):
    assert "\n" not in text
    assert len(text) <= 4096, len(text)
    line = json.loads(text)
    assert json.loads(json.dumps(line)) == line
    for k in CONTRACT_KEYS:
        assert k in line, k
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma")
    for k in ("achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-3 * rf["frac"]
    assert "workload" in line["config"] and "model" not in line["config"]
    return line


def test_compact_line_from_the_round_5_full_record():
    """the very record that did not parse in round 5 (27 KB) gives a compact line with the same headline numbers"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    line = _check(bench.compact_line(full, "/somewhere/bench_full.json"))
    assert abs(line["value"] - full["value"]) <= 1e-6 * full["value"]
    assert line["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert line["cpu_baseline"]["kind"] == "port"
    assert line["summary"]["c2"] == full["summary"]["c2"]
    assert line["full_record"] == "bench_full.json"


def test_compact_line_worst_case_stub():
    """a stub with every optional part at its widest (8 ranks, long strings, a wide summary) still fits"""
    rf = {"bound": "valu", "achieved": 507.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 507.123456789 / 8000.0,
          "traffic": 5.04e9, "hbm_measured_frac": 0.0546, "kernel": "k" * 500, "avg_kernel_ms": 11.69, "launches": 20,
          "concurrent_launches": 2, "algorithmic_bytes_per_launch": 5943296000,
          "valu_issue": {"frac": 0.82, "note": "n" * 2000}, "fp64_vector": {"frac": 0.17}, "traffic_note": "t" * 3000}
    summ = {"unit": "M it/s", "build": "0123456789abcdef"}
    for k in ("c2", "c4", "c5", "tsr1", "tsr3", "held4", "c3_block", "d2", "d3", "held4_tree", "held4_f32"):
        for sfx in ("", "_serial", "_frac", "_parity", "_cpu"):
            summ[k + sfx] = 12.345678
    summ["sweep"] = {str(b): 1.2345 for b in (1, 64, 1024, 4096, 8192, 16384, 65536)}
    out = {"metric": "CHOMP iters/sec, 7-DOF x 100-waypoint", "value": 16931234.5678, "unit": "CHOMP iterations/s", "n_gpus": 8,
           "steps": 20, "warmup": 5, "ms_per_step": 5.95, "scaling": "weak", "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 1000, "runs_per_gpu": 8192, "n_iter": 100, "n_points": 100, "dof": 7, "parallelism": "p" * 60,
                      "knobs": {"a": 1}},
           "value_serial": 12.5e6, "iterations_made": 2.0e6, "iterations_nominal": 2.048e6, "runs_outside_joint_limits": 57,
           "roofline": rf,
           "cpu_baseline": {"value": 34212.7, "unit": "CHOMP iterations/s", "cores": 16, "value_1_core": 2300.1, "host_cores": 16,
                            "kind": "port", "sample": "s" * 800},
           "parity_rel_l2_max_vs_oracle": 3.1e-13, "parity_bound": 1e-6, "parity_runs_checked": 16,
           "parity_ill_conditioned_runs": [{"run": 3}] * 5,
           "per_rank": [{"rank": r, "value": 2.1e6 + r} for r in range(8)], "gather": {"total_s": 1.23}, "backend": "gloo",
           "other_configs": [{"x": "y" * 10000}], "batch_sweep": {"sweep": ["z" * 5000]}, "summary": summ}
    line = _check(bench.compact_line(out, None))
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["counters_say"] == "valu"
    assert len(line["per_rank_value"]) == 8
    assert "other_configs" not in line and "batch_sweep" not in line


def test_compact_line_without_counters_or_cpu_baseline():
    """N > 1 lines carry no cpu baseline (rank 0 at N = 1 only) and a fresh build has no counters: nulls, not a crash"""
    out = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 2, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "scaling": "weak",
           "dtype": "f64", "data": "synthetic", "config": {"workload": "w"},
           "roofline": {"bound": "hbm", "achieved": 80.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.01, "traffic": None,
                        "valu_issue": None, "kernel": "k"},
           "cpu_baseline": None}
    line = _check(bench.compact_line(out))
    assert line["cpu_baseline"] is None and line["roofline"]["traffic"] is None
