# the general joint-limit loop as a function of its own (g2 / g4: config-2 / config-4 builds) against the product
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r05
bash scripts/ab.sh "product g2 product g2" "2" g
bash scripts/ab.sh "product g4 product g4" "4" g4
