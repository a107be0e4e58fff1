cd $GRAFT_REPO_ROOT
bash scripts/ab.sh "f2z f2u f2z f2u" "2" f2u
bash scripts/ab_held4.sh "p6z p6u" 2
