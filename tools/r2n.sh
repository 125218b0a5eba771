cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"
