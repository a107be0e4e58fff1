cd $GRAFT_REPO_ROOT
python scripts/phase_profile_cfg.py 5 2048 30 2>&1 | grep -E "config 5|cost  |FK  "
for v in PASS1 PASS2 SDF JT; do echo "== without $v"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_ablate_$v.so python scripts/phase_profile_cfg.py 5 2048 30 2>&1 | grep -E "config 5|cost  "; done
