cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 60 tools/microbench/div3_check > gpurun_out/q_div3.txt 2>&1 || { cat gpurun_out/q_div3.txt; exit 1; }
cat gpurun_out/q_div3.txt
timeout -k 10 400 python -u tools/r3_dgf.py > gpurun_out/q_dgf.txt 2>&1 || { tail -30 gpurun_out/q_dgf.txt; exit 1; }
tail -1 gpurun_out/q_dgf.txt
timeout -k 10 200 python -u -m pytest tests/test_driver.py tests/test_c_abi_gpu.py -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/q_drv.txt 2>&1 || { tail -30 gpurun_out/q_drv.txt; exit 1; }
tail -1 gpurun_out/q_drv.txt
for cfg in "hd --steps 300" "hd --batch 8 --steps 100" "4k --steps 200"; do
set -- $cfg; wl=$1; shift
timeout -k 10 300 python -u bench.py --workload $wl --no-cpu-baseline --no-also "$@" --dump-kernels gpurun_out/q_kernels_$wl$2$3.json > gpurun_out/q_bench_$wl$2$3.txt 2>&1 || { tail -30 gpurun_out/q_bench_$wl$2$3.txt; exit 1; }
python - "$wl$2$3" <<'PY'
import json, sys
tag = sys.argv[1]
line = open("gpurun_out/q_bench_%s.txt" % tag).read().strip().split("\n")[-1]
j = json.loads(line)
k = json.load(open("gpurun_out/q_kernels_%s.json" % tag))
d = {e["name"]: e["ms"] * 1e3 for e in k}
print(tag, "fps %.1f" % j["value"], "launches", len(k), "sum us %.1f" % sum(d.values()), " dgf_apply %.2f us" % d.get("dgf_apply_fullres", 0), "roofline", j["roofline"].get("frac"), j["roofline"].get("dgf_apply"))
PY
done
