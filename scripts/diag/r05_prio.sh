# wave priorities of the update phase (with the constraint step in it) on the TSR lines: 3 (product's), 1, 0, and no priorities at all
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
AB_EXTRA="--steps 10 --warmup 2 --serial-steps 4" bash scripts/ab.sh "pu3 pu1 pu0 pf0 pu3 pu1 pu0 pf0" "tsr1 tsr3" prio
