cd $GRAFT_REPO_ROOT
XGM_QCOST_LIST=1 timeout 300 python tools/qcost.py --op PHRASE --topk 10 --ref-docs 0 --no-hook-parity 2>&1 | grep QCOST > gpurun_out/r6_qcost_list.txt
timeout 300 python tools/qcost.py --op PHRASE --topk 10 --ref-docs 0 --no-hook-parity 2>&1 | grep QCOST > gpurun_out/r6_qcost_intended.txt
head -30 gpurun_out/r6_qcost_list.txt; echo; head -12 gpurun_out/r6_qcost_intended.txt
