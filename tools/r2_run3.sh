cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -v --timeout 300 --timeout-method thread -k "resnet50" > gpurun_out/r2_pytest3.txt 2>&1 || { tail -40 gpurun_out/r2_pytest3.txt; exit 1; }
tail -5 gpurun_out/r2_pytest3.txt
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --no-also --no-cpu-baseline --workload hd_resnet50 --steps 100 --dump-kernels gpurun_out/r2_k_rn.json > gpurun_out/r2_bench_rn.txt 2>&1 || { tail -20 gpurun_out/r2_bench_rn.txt; exit 1; }
tail -n 1 gpurun_out/r2_bench_rn.txt | cut -c1-300
grep "rvm tune" gpurun_out/r2_bench_rn.txt | tail -80
