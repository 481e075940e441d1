# shorter iteration loop: fp32 + fp16 parity, HD bench, batch-8 bench, in-kernel timeline of the decoder
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --T 2 > gpurun_out/it_check32.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --T 3 > gpurun_out/it_check16io.txt 2>&1 &&
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16fast --T 2 --B 2 > gpurun_out/it_check16.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/it_bench_hd.txt 2>&1 &&
timeout -k 10 300 python -u bench.py --no-cpu-baseline --batch 8 > gpurun_out/it_bench_b8.txt 2>&1 &&
RVM_LIB_VARIANT=stamps timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/stamps_hd.txt 2>&1
echo exit=$?
for f in it_check32 it_check16io it_check16; do tail -n 1 gpurun_out/$f.txt | cut -c1-120; done
for f in gpurun_out/it_bench_hd.txt gpurun_out/it_bench_b8.txt; do tail -n 1 $f | cut -c80-200; done
grep -E "decode|frame" gpurun_out/stamps_hd.txt
