cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
for k in 3; do timeout -k 10 600 python bench.py > gpurun_out/r05/bench_default_again$k.json 2>/dev/null; echo "bench rc $?"; python - <<PY
import json
for l in open("gpurun_out/r05/bench_default_again$k.json"):
    if l.startswith("{"):
        b = json.loads(l); s = b["summary"]; print({k: s[k] for k in s if not k.endswith(("_frac", "_parity", "_cpu")) and k not in ("unit", "sweep")})
PY
done
