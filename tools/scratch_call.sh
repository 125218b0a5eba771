cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SWEEP_STEPS=10
bash tools/sweep.sh "--in-flight 2" X=c2_if2 XGM_NO_EXT_LAUNCH=1
bash tools/sweep.sh "--in-flight 3" X=c2_if3 XGM_NO_EXT_LAUNCH=1
bash tools/sweep.sh "--in-flight 2" X=c2_if2
bash tools/sweep.sh "--in-flight 3" X=c2_if3
export SWEEP_STEPS=4
bash tools/sweep.sh "--op OR --terms 5 --topk 100 --in-flight 2" X=c3_if2
bash tools/sweep.sh "--op OR --terms 5 --topk 100 --in-flight 3" X=c3_if3
bash tools/sweep.sh "--op PHRASE --topk 10 --in-flight 2" X=c5_if2
bash tools/sweep.sh "--op PHRASE --topk 10 --in-flight 3" X=c5_if3
