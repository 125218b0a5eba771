cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
XGM_LIB_PATH=$GRAFT_REPO_ROOT/xapiand_amd/csrc/ab/libxgm_mt.so timeout 300 python tools/phase.py --no-latency --threads 0 2>&1 | grep "MERGE PHASES"
