R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/b
timeout -k 10 400 python -u bench.py --dump-kernels gpurun_out/b/k_hd.json > gpurun_out/b/hd.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload 4k --dump-kernels gpurun_out/b/k_4k.json > gpurun_out/b/4k.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --workload hd_resnet50 --dump-kernels gpurun_out/b/k_rn.json > gpurun_out/b/rn.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --dtype fp16fast --dump-kernels gpurun_out/b/k_hd_fast.json > gpurun_out/b/hd_fast.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --dtype fp16fast --workload 4k > gpurun_out/b/4k_fast.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --dtype fp16fast --workload hd_resnet50 --dump-kernels gpurun_out/b/k_rn_fast.json > gpurun_out/b/rn_fast.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --batch 8 --steps 100 > gpurun_out/b/hd_b8.txt 2>&1
echo exit=$?
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b/prof_hd -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/b/prof_stdout.txt 2>&1
echo prof_exit=$?
cd $R; for f in hd 4k rn hd_fast 4k_fast rn_fast hd_b8; do echo $f; tail -n 1 gpurun_out/b/$f.txt | cut -c80-175; done
