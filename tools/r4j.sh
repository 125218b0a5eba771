cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_run.sh "--op PHRASE --topk 10 --steps 6 --warmup 1 --no-other-configs" default he4 he8 he16 pw2 pw4 he8pw4
