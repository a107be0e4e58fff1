# the tile loop as one function per iteration with FK inlined in it (tl1) against FK as a call per tile (tl0): config-2 builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "tl0 tl1 tl0 tl1" "2" tl
