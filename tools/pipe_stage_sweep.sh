# Frame pipeline: encoder stages 1 / 2 / 3 on the bench workloads (WL as in tools/launch_times.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ${WL:-hd 4k b8 rn}; do
case $wl in
  hd) args="--steps 300" ;;
  4k) args="--workload 4k --steps 200" ;;
  b8) args="--batch 8 --steps 100" ;;
  rn) args="--workload hd_resnet50 --steps 100" ;;
  c12) args="--chunk 12 --steps 50" ;;
esac
for st in ${STAGES:-1 2 3}; do
RVM_PIPE_STAGES=$st timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also $args --pipeline on > gpurun_out/pipe_sweep.txt 2>&1 || { tail -30 gpurun_out/pipe_sweep.txt; exit 1; }
python - $wl $st <<'PY'
import json, sys
line = json.loads(open("gpurun_out/pipe_sweep.txt").read().strip().splitlines()[-1])
print("%s stages=%s: %.1f frames/s (%.1f us/step)  one stream %.1f  mem %d MB" % (sys.argv[1], sys.argv[2], line["value"], line["ms_per_step"] * 1e3, line["fps_pipeline_off"], line["memory_bytes"]["total"] >> 20))
PY
done; done
