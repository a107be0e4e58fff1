# with the lean update phase in: the FK phase inside the kernel function (if1), the 16-lane cost pass inside it (ic2), against calls (b2)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "b2 if1 ic2 b2 if1 ic2" "2" inl
