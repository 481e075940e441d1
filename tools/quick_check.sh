# One pass on the GPU box after a kernel change: quick parity (fp32 small with odd sizes, fp16 HD, fp32 batch 2), then tools/launch_times.sh.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ -n "$MB" ]; then timeout -k 10 120 tools/microbench/$MB > gpurun_out/mb_$MB.txt 2>&1 || { tail gpurun_out/mb_$MB.txt; exit 1; }; cat gpurun_out/mb_$MB.txt; fi
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.25 --T 2 > gpurun_out/q_fp32_small.txt 2>&1 || { tail -30 gpurun_out/q_fp32_small.txt; exit 1; }
tail -1 gpurun_out/q_fp32_small.txt | cut -c1-400
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --H 1080 --W 1920 --ds 0.25 --T 2 > gpurun_out/q_fp16_hd.txt 2>&1 || { tail -30 gpurun_out/q_fp16_hd.txt; exit 1; }
tail -1 gpurun_out/q_fp16_hd.txt | cut -c1-400
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2 > gpurun_out/q_fp32_b2.txt 2>&1 || { tail -30 gpurun_out/q_fp32_b2.txt; exit 1; }
tail -1 gpurun_out/q_fp32_b2.txt | cut -c1-400
bash tools/launch_times.sh
