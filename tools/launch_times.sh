# Per-launch timings of the current build: HD / 4K single stream, 8 batched HD streams (WL = space-separated subset of: hd 4k b8 rn).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ${WL:-hd 4k b8}; do
case $wl in
  hd) args="--steps 300" ;;
  4k) args="--workload 4k --steps 200" ;;
  b8) args="--batch 8 --steps 100" ;;
  rn) args="--workload hd_resnet50 --steps 100" ;;
esac
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also $args --dump-kernels gpurun_out/now_kernels_$wl.json > gpurun_out/now_bench_$wl.txt 2>&1 || { tail -30 gpurun_out/now_bench_$wl.txt; exit 1; }
python - $wl <<'PY'
import json, sys
wl = sys.argv[1]
line = json.loads(open("gpurun_out/now_bench_%s.txt" % wl).read().strip().splitlines()[-1])
k = json.load(open("gpurun_out/now_kernels_%s.json" % wl))
print("%s: %.1f frames/s  %.4f ms/step  launches %d  sum %.1f us" % (wl, line["value"], line["ms_per_step"], len(k), 1e3 * sum(e["ms"] for e in k)))
print("   " + "  ".join("%s %.1f" % (e["name"], e["ms"] * 1e3) for e in sorted(k, key=lambda e: -e["ms"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]))
PY
done
