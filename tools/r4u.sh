cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_ortim.so timeout 300 python tools/phase.py --op OR --terms 5 --topk 100 --no-other-configs --no-latency --threads 0 2>&1 | grep -E "PHASES|value" | cut -c1-900
