cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/qcost.py --op PHRASE --topk 10 2>/dev/null | grep QCOST > gpurun_out/r04e_qcost_phrase.txt; cat gpurun_out/r04e_qcost_phrase.txt
timeout 300 python tools/qcost.py --op OR --terms 5 --topk 100 2>/dev/null | grep QCOST > gpurun_out/r04e_qcost_or.txt; cat gpurun_out/r04e_qcost_or.txt
timeout 300 python tools/qcost.py 2>/dev/null | grep QCOST > gpurun_out/r04e_qcost_and.txt; cat gpurun_out/r04e_qcost_and.txt
