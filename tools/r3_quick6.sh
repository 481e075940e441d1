cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 200 "$@" > gpurun_out/$tag.txt 2>&1 || { tail -30 gpurun_out/$tag.txt; exit 1; }; echo "$tag: $(tail -1 gpurun_out/$tag.txt | cut -c 10-130)"; }
export RVM_AUTOTUNE=0
run m_fp32 python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2
for t in 2,8 4,4 4,8 2,16 8,4; do
RVM_MBF_TILE=$t run m_fp32_$t python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.5 --T 2
done
RVM_MB1_FOLD=0 run m_fp32_off python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2
run m_f16io python -u tools/gpu_check.py --dtype fp16 --H 1080 --W 1920 --ds 0.25 --T 2
run m_fp16fast python -u tools/gpu_check.py --dtype fp16fast --H 150 --W 201 --ds 0.8 --T 2 --B 2
run m_bf16 python -u tools/gpu_check.py --dtype bf16 --H 97 --W 131 --ds 1 --T 2
unset RVM_AUTOTUNE
RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd.txt; exit 1; }
grep -E "frame |mb[12]\.|downsample" gpurun_out/q_stamps_hd.txt | grep -v "rvm tune" | cut -c1-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --steps 300 --dump-kernels gpurun_out/q_kernels_hd.json > gpurun_out/q_bench_hd.txt 2>&1 || { tail -30 gpurun_out/q_bench_hd.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_hd.txt | cut -c80-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --batch 8 --steps 100 --dump-kernels gpurun_out/q_kernels_b8.json > gpurun_out/q_bench_b8.txt 2>&1 || { tail -30 gpurun_out/q_bench_b8.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_b8.txt | cut -c80-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --workload 4k --steps 200 > gpurun_out/q_bench_4k.txt 2>&1 || { tail -30 gpurun_out/q_bench_4k.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_4k.txt | cut -c80-200
python - <<'PY'
import json
k = json.load(open("gpurun_out/q_kernels_b8.json"))
print("b8 launches", len(k), "sum ms %.3f" % sum(e["ms"] for e in k))
for e in k[:4]: print("   %-26s %.2f us" % (e["name"], e["ms"] * 1e3))
PY
