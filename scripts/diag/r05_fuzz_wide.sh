cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
ORC_RANDOM_SCENES=300 timeout -k 10 600 python -m pytest tests/test_gpu_sdf_fuzz.py -q > gpurun_out/r05/sdf_fuzz_wide.txt 2>&1; echo "sdf rc $?"; tail -n 1 gpurun_out/r05/sdf_fuzz_wide.txt
ORC_COMMAND_FUZZ=40000 ORC_COMMAND_FUZZ_SEED=777 timeout -k 10 600 python -m pytest tests/test_gpu_command_fuzz.py -q > gpurun_out/r05/command_fuzz_wide.txt 2>&1; echo "fuzz rc $?"; tail -n 1 gpurun_out/r05/command_fuzz_wide.txt
