cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_grabbed.py tests/test_gpu_held4.py -x -q -s > gpurun_out/r05/pairs2_test.txt 2>&1; echo "test rc $?" >> gpurun_out/r05/pairs2_test.txt
tail -15 gpurun_out/r05/pairs2_test.txt
timeout -k 10 300 python bench.py --config held4 --steps 16 --warmup 2 > gpurun_out/r05/bench_held4_a.json 2> gpurun_out/r05/bench_held4_a.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_held4_a.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "value_serial", "ms_per_step", "parity_rel_l2_max_vs_oracle", "runs_outside_joint_limits")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["config"]["knobs"])
PY
