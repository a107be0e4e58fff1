cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
export ORC_LIB=$GRAFT_REPO_ROOT/or_cdchomp_amd/liborcdchomp_var_p6.so
timeout -k 10 300 python -m pytest tests/test_gpu_grabbed.py -x -q -k "four_sphere_body and not fp32" > gpurun_out/r05/pairs1_test.txt 2>&1; echo "test rc $?" >> gpurun_out/r05/pairs1_test.txt
tail -5 gpurun_out/r05/pairs1_test.txt
timeout -k 10 200 python scripts/phase_profile_held4.py > gpurun_out/r05/pairs1_phase_w3.txt 2>&1; tail -9 gpurun_out/r05/pairs1_phase_w3.txt
WGS_PER_CU=4 timeout -k 10 200 python scripts/phase_profile_held4.py > gpurun_out/r05/pairs1_phase_w4.txt 2>&1; tail -9 gpurun_out/r05/pairs1_phase_w4.txt
