cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/kstat.sh or2 --no-other-configs --op OR --terms 5 --topk 100 > gpurun_out/r5e_kstat.txt 2>&1; cat gpurun_out/r5e_kstat.txt
