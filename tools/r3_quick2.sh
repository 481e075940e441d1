cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.25 --T 2 > gpurun_out/q_fp32_small.txt 2>&1 || { tail -30 gpurun_out/q_fp32_small.txt; exit 1; }
tail -1 gpurun_out/q_fp32_small.txt
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --H 1080 --W 1920 --ds 0.25 --T 2 > gpurun_out/q_fp16_hd.txt 2>&1 || { tail -30 gpurun_out/q_fp16_hd.txt; exit 1; }
tail -1 gpurun_out/q_fp16_hd.txt
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16fast --H 150 --W 201 --ds 0.8 --T 2 --B 2 > gpurun_out/q_fp16fast_b2.txt 2>&1 || { tail -30 gpurun_out/q_fp16fast_b2.txt; exit 1; }
tail -1 gpurun_out/q_fp16fast_b2.txt
RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd.txt; exit 1; }
grep -E "frame |fused" gpurun_out/q_stamps_hd.txt | grep -v wres
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --steps 300 --dump-kernels gpurun_out/q_kernels_hd.json > gpurun_out/q_bench_hd.txt 2>&1 || { tail -30 gpurun_out/q_bench_hd.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_hd.txt | cut -c80-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --batch 8 --steps 100 --dump-kernels gpurun_out/q_kernels_b8.json > gpurun_out/q_bench_b8.txt 2>&1 || { tail -30 gpurun_out/q_bench_b8.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_b8.txt | cut -c80-200
python - <<'PY'
import json
k = json.load(open("gpurun_out/q_kernels_b8.json"))
print("b8 launches", len(k), "sum ms %.3f" % sum(e["ms"] for e in k))
for e in k:
    if "fused" in e["name"]: print("   %-22s %.2f us" % (e["name"], e["ms"] * 1e3))
PY
