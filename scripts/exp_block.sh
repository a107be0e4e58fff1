cd $GRAFT_REPO_ROOT
for blk in 256 192; do
  ORC_BLOCK_THREADS=$blk python scripts/quick_bench.py 1024,4096,16384 6 2>&1 | tail -3
  ORC_BLOCK_THREADS=$blk NSTREAMS=3 python scripts/quick_bench.py 1024 12 2>&1 | tail -1
done
python -m pytest tests -q -m gpu 2>&1 | tail -30
