# Full GPU pass (GPU box): every -m gpu test, the allocation-free entry point's test alone in a fresh process, the default bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/r2_pytest.txt 2>&1 || { tail -40 gpurun_out/r2_pytest.txt; exit 1; }
tail -5 gpurun_out/r2_pytest.txt
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -x -v --timeout 100 --timeout-method thread > gpurun_out/r2_noalloc.txt 2>&1 || { tail -40 gpurun_out/r2_noalloc.txt; exit 1; }
tail -4 gpurun_out/r2_noalloc.txt
timeout -k 10 400 python -u bench.py > gpurun_out/r2_bench.txt 2>&1 || { tail -20 gpurun_out/r2_bench.txt; exit 1; }
tail -n 1 gpurun_out/r2_bench.txt | cut -c1-1500
