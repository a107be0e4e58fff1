cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ORC_DEBUG_PLAN=1 timeout -k 10 300 python -m pytest tests/test_gpu_regressions.py -q -s 2>&1 | grep -v "^$" | tail -n 8 | cut -c1-250
