cd $GRAFT_REPO_ROOT
export SIZES=1000000 CONFIGS=1024:0:4
sh tools/ablate.sh "128 0 2 12 30" 2>&1 | grep -v "^$"
