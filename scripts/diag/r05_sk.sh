# wavefronts without a waypoint in a tile skip the FK call (sk7 / sk2) against calling it (sk70 / sk20): TSR-only and config-2 builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
AB_EXTRA="--steps 10 --warmup 2 --serial-steps 4" bash scripts/ab.sh "sk70 sk7 sk70 sk7" "tsr1 tsr3" sk7
bash scripts/ab.sh "sk20 sk2 sk20 sk2" "2" sk2
