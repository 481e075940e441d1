cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --T 2 > gpurun_out/r3_mb32.txt 2>&1 &&
timeout -k 10 300 python -u tools/gpu_check.py --dtype fp32 --T 2 --variant resnet50 > gpurun_out/r3_rn32.txt 2>&1 &&
timeout -k 10 300 python -u tools/gpu_check.py --dtype fp16fast --T 2 --variant resnet50 > gpurun_out/r3_rn16.txt 2>&1
echo exit=$?
tail -n 1 gpurun_out/r3_mb32.txt | cut -c1-800; tail -n 25 gpurun_out/r3_rn32.txt | cut -c1-800; tail -n 1 gpurun_out/r3_rn16.txt | cut -c1-800
