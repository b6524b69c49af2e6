cd $GRAFT_REPO_ROOT
GTARS_TOK_QPT=8 GTARS_TOK_TPB=512 timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
SIZES=16000000,64000000,256000000 CONFIGS=512:0:4,512:0:8,1024:0:8 python tools/kbench.py
