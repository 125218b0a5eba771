cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab
XGM_LIB_PATH=$AB/libxgm_clk2.so XGM_QCOST_CLOCKS=1 timeout 300 python tools/qcost.py --op PHRASE --topk 10 2>&1 | grep QCOST | head -22
