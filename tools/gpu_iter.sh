# quick iteration loop on the GPU box: parity first, then timing of the HD workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --T 2 > gpurun_out/it_check32.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --T 3 > gpurun_out/it_check16io.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --T 2 --H 1080 --W 1920 > gpurun_out/it_check16io_hd.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16fast --T 2 --B 2 > gpurun_out/it_check16.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --dump-kernels gpurun_out/it_kernels_hd.json > gpurun_out/it_bench_hd.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --dtype fp16fast --dump-kernels gpurun_out/it_kernels_hd_fast.json > gpurun_out/it_bench_hd_fast.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload 4k > gpurun_out/it_bench_4k.txt 2>&1
echo exit=$?
for f in it_check32 it_check16io it_check16io_hd it_check16; do tail -n 1 gpurun_out/$f.txt | cut -c1-700; done
for f in gpurun_out/it_bench_hd.txt gpurun_out/it_bench_hd_fast.txt gpurun_out/it_bench_4k.txt; do tail -n 1 $f | cut -c80-200; done
