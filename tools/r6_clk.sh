cd $GRAFT_REPO_ROOT
XGM_LIB_PATH=$PWD/xapiand_amd/csrc/ab/libxgm_clk2.so XGM_QCOST_CLOCKS=1 XGM_QCOST_LIST=1 timeout 300 python tools/qcost.py --op PHRASE --topk 10 --ref-docs 0 --no-hook-parity 2>&1 | grep QCOST > gpurun_out/r6_qcost_clk.txt
head -24 gpurun_out/r6_qcost_clk.txt
