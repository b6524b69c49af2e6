cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
export SIZES=64000000 CONFIGS=512:0:4
sh tools/ablate.sh "128" 2>&1 | grep -v "^$"
SIZES=1000000,4000000,16000000,64000000,256000000 CONFIGS=512:0:4,1024:0:4 sh tools/ablate.sh "0" 2>&1 | grep -v "^$"
