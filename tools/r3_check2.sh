cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
#PART=b bash tools/gpu_suite.sh || exit 1
timeout -k 10 300 python -u -m pytest tests/test_driver.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3_driver.txt 2>&1 || { tail -30 gpurun_out/r3_driver.txt; exit 1; }
tail -3 gpurun_out/r3_driver.txt
( time timeout -k 10 600 python -u bench.py --dump-kernels gpurun_out/r3_kernels_hd.json > gpurun_out/r3_bench.txt 2> gpurun_out/r3_bench_err.txt ) 2>&1 | grep real
tail -n 1 gpurun_out/r3_bench.txt | cut -c1-400; tail -5 gpurun_out/r3_bench_err.txt
