# usage: tools/calib.sh <tag>: FETCH_SIZE / WRITE_SIZE of tools/fetch_calib's known-traffic kernels (one MI355X, ~1 min)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_calib_$c -- tools/fetch_calib > gpurun_out/${tag}_calib_$c.log 2>&1
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_calib_stats -- tools/fetch_calib > gpurun_out/${tag}_calib_known.json 2>gpurun_out/${tag}_calib_stats.err
python tools/pmc_parse.py gpurun_out/${tag}_calib_FETCH_SIZE gpurun_out/${tag}_calib_WRITE_SIZE | tee gpurun_out/${tag}_calib_pmc.txt
python tools/calib_report.py gpurun_out/${tag}_calib_known.json gpurun_out/${tag}_calib_pmc.txt gpurun_out/${tag}_calib_stats | tee gpurun_out/${tag}_fetch_calib.txt
