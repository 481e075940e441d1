cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ${WL:-hd 4k b8 rn}; do
case $wl in
  hd) args="--steps 300" ;;
  4k) args="--workload 4k --steps 200" ;;
  b8) args="--batch 8 --steps 100" ;;
  rn) args="--workload hd_resnet50 --steps 100" ;;
  c12) args="--chunk 12 --steps 50" ;;
esac
for mode in eager graph; do
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also $args --pipeline on --mode $mode > gpurun_out/pipe_bench_${wl}_$mode.txt 2>&1 || { tail -30 gpurun_out/pipe_bench_${wl}_$mode.txt; exit 1; }
python - $wl $mode <<'PY'
import json, sys
wl, mode = sys.argv[1:3]
line = json.loads(open("gpurun_out/pipe_bench_%s_%s.txt" % (wl, mode)).read().strip().splitlines()[-1])
print("%s %s: pipeline on %.1f frames/s (%.1f us/step)  sync-every-frame %.1f  single-src %.1f" % (wl, mode, line["value"], line["ms_per_step"] * 1e3, line["fps_sync_every_frame"], line["fps_single_src"]))
PY
done; done
