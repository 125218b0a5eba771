cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_run.sh "--op OR --terms 5 --topk 100 --steps 4 --warmup 1 --no-other-configs" default orw2c orw3a orw3b orw3g2
XGM_ORW_UNITS=12288 bash tools/ab_run.sh "--op OR --terms 5 --topk 100 --steps 4 --warmup 1 --no-other-configs" default orw3a
