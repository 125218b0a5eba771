# round 6: the hook's combined launches on the MI355X, then the default bench line (hook parity from 64 threads included)
python -m pytest tests/test_gpu_hook_b1.py -x -q -m gpu -k "share_launches or many_threads" > gpurun_out/r6_hook_pytest.log 2>&1
tail -5 gpurun_out/r6_hook_pytest.log
python bench.py > gpurun_out/r6_default_bench.json 2> gpurun_out/r6_default_bench.err
tail -c 600 gpurun_out/r6_default_bench.json
