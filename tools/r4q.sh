cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_clk.so XGM_QCOST_CLOCKS=1 timeout 300 python tools/qcost.py --op PHRASE --topk 10 2>&1 | grep QCOST
XGM_UNITS_PER_QUERY=96 XGM_TARGET_UNITS=24576 XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_clk.so XGM_QCOST_CLOCKS=1 timeout 300 python tools/qcost.py --op PHRASE --topk 10 2>&1 | grep QCOST | head -12
