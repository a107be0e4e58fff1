cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 300 python3 scripts/phase_profile_cfg.py 5 2>&1 | grep -v "orc placement" | tail -14
