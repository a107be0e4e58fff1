# backend flags on the config-2 kernels: amdgpu trackers in the scheduler, regclass priority in the allocator, no scalar IR passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "base trk gp sir base trk gp sir" "2" flags
