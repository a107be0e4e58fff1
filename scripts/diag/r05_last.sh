cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout -k 10 600 python -m pytest tests/ -q -m gpu > gpurun_out/r05/gputests_final.txt 2>&1; echo "tests rc $?"; tail -n 1 gpurun_out/r05/gputests_final.txt
timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_last.json 2>/dev/null; echo "bench rc $?"; python - <<'PY'
import json
for l in open("gpurun_out/r05/bench_default_last.json"):
    if l.startswith("{"):
        b = json.loads(l); print(json.dumps(b["summary"])[:700]); print("traffic", b["roofline"]["traffic"], "valu", b["roofline"]["valu_issue"]["frac"] if b["roofline"]["valu_issue"] else None)
PY
