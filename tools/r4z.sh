cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab
XGM_LIB_PATH=$AB/libxgm_clk.so XGM_QCOST_CLOCKS=1 timeout 300 python tools/qcost.py --op AND --terms 3 --topk 10 2>&1 | grep QCOST | grep -v "phases of" | tail -8
timeout 300 python tools/qcost.py --op AND --terms 3 --topk 10 2>&1 | grep QCOST | grep -v "phases of" | awk '{print $0}' | grep -E "all-dense (True|False)" | awk '{c[$NF]++; s[$NF]+=$5} END{for(k in c) print "top-24 by kind", k, c[k], s[k]}'
