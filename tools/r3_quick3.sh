cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 200 "$@" > gpurun_out/$tag.txt 2>&1 || { tail -30 gpurun_out/$tag.txt; exit 1; }; echo "$tag: $(tail -1 gpurun_out/$tag.txt | cut -c 150-420)"; }
run g_fp32_small python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.25 --T 2
run g_fp32_ds1 python -u tools/gpu_check.py --dtype fp32 --H 70 --W 90 --ds 1 --T 3 --B 2
RVM_GRU_TILE=4,8 run g_fp32_t48 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.5 --T 2
RVM_GRU_TILE=8,16 run g_fp32_t816 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.5 --T 2
RVM_GRU_TILE=8,8 run g_fp32_t88 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.5 --T 2
RVM_GRU_TILE=4,16 run g_f16io_t416 python -u tools/gpu_check.py --dtype fp16 --H 122 --W 218 --ds 0.5 --T 2
RVM_GRU_FUSE=0 run g_fp32_off python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.5 --T 2
run g_fp16_hd python -u tools/gpu_check.py --dtype fp16 --H 1080 --W 1920 --ds 0.25 --T 2
run g_bf16 python -u tools/gpu_check.py --dtype bf16 --H 300 --W 400 --ds 0.5 --T 3
run g_fp16fast_b2 python -u tools/gpu_check.py --dtype fp16fast --H 150 --W 201 --ds 0.8 --T 2 --B 2
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
    if "fused" in e["name"] or "gru" in e["name"]: print("   %-22s %.2f us" % (e["name"], e["ms"] * 1e3))
PY
