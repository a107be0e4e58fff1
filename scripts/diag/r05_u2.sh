cd $GRAFT_REPO_ROOT
bash scripts/ab.sh "c4z c4y c4z c4y" "4" c4y
bash scripts/ab.sh "f2z f2y f2z f2y" "2" f2y
