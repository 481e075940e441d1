cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 200 "$@" > gpurun_out/$tag.txt 2>&1 || { tail -30 gpurun_out/$tag.txt; exit 1; }; echo "$tag: $(tail -1 gpurun_out/$tag.txt | cut -c 150-420)"; }
for t in 0 1 2 3 4 5 6 7 8 9 10 11; do
RVM_CONV_REG_VAR=$t run c_fp32_v$t python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2
done
RVM_CONV_REG_VAR=4 run c_f16io python -u tools/gpu_check.py --dtype fp16 --H 540 --W 960 --ds 0.5 --T 2
RVM_CONV_REG_VAR=11 run c_f16io11 python -u tools/gpu_check.py --dtype fp16 --H 540 --W 960 --ds 0.5 --T 2
RVM_CONV_REG_VAR=0 run c_fp16fast python -u tools/gpu_check.py --dtype fp16fast --H 122 --W 218 --ds 0.5 --T 2
RVM_CONV_REG=0 run c_off python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2
run c_default_hd python -u tools/gpu_check.py --dtype fp16 --H 1080 --W 1920 --ds 0.25 --T 2
RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd.txt; exit 1; }
grep -E "frame |decode[0-9].conv|conv_cat" gpurun_out/q_stamps_hd.txt | grep -v wres | cut -c1-200
for t in 0 1 2 3 4 5 6 7 8 9 10 11; do
RVM_CONV_REG_VAR=$t timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd_t$t.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd_t$t.txt; exit 1; }
echo "variant $t"; grep -E "frame |decode[0-9].conv " gpurun_out/q_stamps_hd_t$t.txt | cut -c1-100
done
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --steps 300 --dump-kernels gpurun_out/q_kernels_hd.json > gpurun_out/q_bench_hd.txt 2>&1 || { tail -30 gpurun_out/q_bench_hd.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_hd.txt | cut -c80-200
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --batch 8 --steps 100 --dump-kernels gpurun_out/q_kernels_b8.json > gpurun_out/q_bench_b8.txt 2>&1 || { tail -30 gpurun_out/q_bench_b8.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_b8.txt | cut -c80-200
grep "rvm tune\] decode[0-9].conv" gpurun_out/q_bench_b8.txt | head -90
python - <<'PY'
import json
k = json.load(open("gpurun_out/q_kernels_b8.json"))
print("b8 launches", len(k), "sum ms %.3f" % sum(e["ms"] for e in k))
for e in k:
    if "decode" in e["name"]: print("   %-26s %.2f us" % (e["name"], e["ms"] * 1e3))
PY
