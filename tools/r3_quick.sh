# Quick GPU pass of the round-3 work in progress: parity of the fused inverted-residual kernel, then per-launch timings.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --H 122 --W 218 --ds 0.25 --T 2 > gpurun_out/q_fp32_small.txt 2>&1 || { tail -30 gpurun_out/q_fp32_small.txt; exit 1; }
tail -1 gpurun_out/q_fp32_small.txt
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp16 --H 1080 --W 1920 --ds 0.25 --T 2 > gpurun_out/q_fp16_hd.txt 2>&1 || { tail -30 gpurun_out/q_fp16_hd.txt; exit 1; }
tail -1 gpurun_out/q_fp16_hd.txt
timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2 > gpurun_out/q_fp32_b2.txt 2>&1 || { tail -30 gpurun_out/q_fp32_b2.txt; exit 1; }
tail -1 gpurun_out/q_fp32_b2.txt
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --steps 300 --dump-kernels gpurun_out/q_kernels_hd.json > gpurun_out/q_bench_hd.txt 2>&1 || { tail -30 gpurun_out/q_bench_hd.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_hd.txt | cut -c1-200
RVM_MB_FUSE=0 timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --steps 300 --dump-kernels gpurun_out/q_kernels_hd_nofuse.json > gpurun_out/q_bench_hd_nofuse.txt 2>&1 || { tail -30 gpurun_out/q_bench_hd_nofuse.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_hd_nofuse.txt | cut -c1-200
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --batch 8 --steps 100 --dump-kernels gpurun_out/q_kernels_b8.json > gpurun_out/q_bench_b8.txt 2>&1 || { tail -30 gpurun_out/q_bench_b8.txt; exit 1; }
tail -n 1 gpurun_out/q_bench_b8.txt | cut -c1-200
python - <<'PY'
import json
for f in ("q_kernels_hd", "q_kernels_hd_nofuse", "q_kernels_b8"):
    k = json.load(open("gpurun_out/%s.json" % f))
    print(f, "launches", len(k), "sum ms %.3f" % sum(e["ms"] for e in k))
    for e in k:
        if e["name"].startswith("mb") and int(e["name"].split(".")[0][2:]) in (2, 3, 7, 8, 9, 10):
            print("   %-22s %.2f us" % (e["name"], e["ms"] * 1e3))
PY
