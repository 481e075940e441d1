cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RVM_VERBOSE=1 timeout -k 10 400 python -u bench.py --no-cpu-baseline --no-also --workload hd_resnet50 --steps 100 --dump-kernels gpurun_out/q_kernels_rn.json > gpurun_out/q_bench_rn.txt 2>&1 || { tail -30 gpurun_out/q_bench_rn.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_rn.txt | cut -c80-200
grep -E "conv_cat|gru.fused" gpurun_out/q_bench_rn.txt | sort -u | head
python - <<'PY'
import json
k = json.load(open("gpurun_out/q_kernels_rn.json"))
print("rn launches", len(k), "sum ms %.3f" % sum(e["ms"] for e in k))
for e in sorted(k, key=lambda e: -e["ms"])[:30]: print("   %-26s %.2f us" % (e["name"], e["ms"] * 1e3))
PY
timeout -k 10 300 python -u bench.py --workload 4k --driver-pipeline > gpurun_out/q_bench_driver.txt 2>&1 || { tail -30 gpurun_out/q_bench_driver.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_driver.txt | cut -c1-1500
