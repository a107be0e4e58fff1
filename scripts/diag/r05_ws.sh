# scan rounds: DPP wave shifts and the kept winner's column (ws) against neither (ws0) and the kept column alone (ws1); config 4 builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "ws0 ws1 ws ws0 ws1 ws" "4" ws
for v in ws0 ws; do echo "== $v"; ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_$v.so timeout -k 10 200 python scripts/phase_profile_cfg.py 4 2>&1 | grep "joint limits\|config 4:"; done
