# pipelined pw_gemm_lds: forced parity, then resnet50 / HD benches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -k "lds_tiled" -x -v --timeout 300 --timeout-method thread > gpurun_out/pw_pytest.txt 2>&1 || { tail -40 gpurun_out/pw_pytest.txt; exit 1; }
tail -6 gpurun_out/pw_pytest.txt
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --workload hd_resnet50 --steps 200 --no-also > gpurun_out/pw_bench_rn.txt 2>&1 || { tail -20 gpurun_out/pw_bench_rn.txt; exit 1; }
tail -n 1 gpurun_out/pw_bench_rn.txt | cut -c1-400
grep "rvm tune" gpurun_out/pw_bench_rn.txt > gpurun_out/pw_tune_rn.txt; wc -l gpurun_out/pw_tune_rn.txt
