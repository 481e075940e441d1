mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --T 3 > gpurun_out/check16.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --T 2 --B 2 --ds 1 --H 70 --W 90 > gpurun_out/check32_ds1.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --T 2 --refiner fast_guided_filter > gpurun_out/check32_fast.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --T 3 --graph > gpurun_out/check32_graph.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --mode eager --no-cpu-baseline --dump-kernels gpurun_out/kernels_hd_eager.json > gpurun_out/bench_hd_eager.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --mode graph --no-cpu-baseline > gpurun_out/bench_hd_graph.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --mode graph --workload 4k --no-cpu-baseline --dump-kernels gpurun_out/kernels_4k.json > gpurun_out/bench_4k_graph.txt 2>&1
echo exit=$?
tail -3 gpurun_out/check16.txt gpurun_out/check32_ds1.txt gpurun_out/check32_fast.txt gpurun_out/check32_graph.txt gpurun_out/bench_*.txt
