cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_clk.so XGM_QCOST_CLOCKS=1 timeout 300 python tools/qcost.py --op AND --terms 3 --topk 10 2>&1 | grep QCOST | head -40
