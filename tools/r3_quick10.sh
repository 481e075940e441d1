cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 200 "$@" > gpurun_out/$tag.txt 2>&1 || { tail -30 gpurun_out/$tag.txt; exit 1; }; echo "$tag: $(tail -1 gpurun_out/$tag.txt | cut -c 10-130)"; }
RVM_AUTOTUNE=0 run m_fp32 python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2
RVM_AUTOTUNE=0 RVM_MBF_TILE=2,16 run m_fp32_216 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.5 --T 2
RVM_AUTOTUNE=0 run m_fp16fast python -u tools/gpu_check.py --dtype fp16fast --H 150 --W 201 --ds 0.8 --T 2 --B 2
timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd.txt; exit 1; }
grep -E "frame |mb2\." gpurun_out/q_stamps_hd.txt | cut -c1-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --steps 300 > gpurun_out/q_bench_hd.txt 2>&1 || { tail -30 gpurun_out/q_bench_hd.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_hd.txt | cut -c80-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --batch 8 --steps 100 --dump-kernels gpurun_out/q_kernels_b8.json > gpurun_out/q_bench_b8.txt 2>&1 || { tail -30 gpurun_out/q_bench_b8.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_b8.txt | cut -c80-200
python - <<'PY'
import json
k = json.load(open("gpurun_out/q_kernels_b8.json"))
print("b8 sum ms %.3f" % sum(e["ms"] for e in k), [(e["name"], round(e["ms"]*1e3,1)) for e in k[:3]])
PY
