cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -k "chunked_decoder" -x -v --timeout 300 --timeout-method thread > gpurun_out/alt_pytest.txt 2>&1 || { tail -40 gpurun_out/alt_pytest.txt; exit 1; }
tail -3 gpurun_out/alt_pytest.txt
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --steps 300 --no-also --no-cpu-baseline > gpurun_out/hd_tune.txt 2>&1 || { tail -20 gpurun_out/hd_tune.txt; exit 1; }
tail -n 1 gpurun_out/hd_tune.txt | cut -c1-200
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --batch 8 --steps 50 --no-also --no-cpu-baseline > gpurun_out/b8_tune.txt 2>&1 || { tail -20 gpurun_out/b8_tune.txt; exit 1; }
tail -n 1 gpurun_out/b8_tune.txt | cut -c1-200
