# Frame pipeline: its tests, then the bench line with the pipeline off / on (WL as in tools/launch_times.sh).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_frame_pipeline.py -x -v --timeout 300 --timeout-method thread > gpurun_out/pipe_pytest.txt 2>&1 || { tail -40 gpurun_out/pipe_pytest.txt; exit 1; }
tail -12 gpurun_out/pipe_pytest.txt
for wl in ${WL:-hd 4k b8}; do
case $wl in
  hd) args="--steps 300" ;;
  4k) args="--workload 4k --steps 200" ;;
  b8) args="--batch 8 --steps 100" ;;
  rn) args="--workload hd_resnet50 --steps 100" ;;
  c12) args="--chunk 12 --steps 50" ;;
esac
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also $args --pipeline on > gpurun_out/pipe_bench_$wl.txt 2>&1 || { tail -30 gpurun_out/pipe_bench_$wl.txt; exit 1; }
python - $wl <<'PY'
import json, sys
wl = sys.argv[1]
line = json.loads(open("gpurun_out/pipe_bench_%s.txt" % wl).read().strip().splitlines()[-1])
print("%s: pipeline on %.1f frames/s (%.4f ms/step)  off %.1f  sync-every-frame %.1f  single-src %.1f  mem %s" % (
    wl, line["value"], line["ms_per_step"], line.get("fps_pipeline_off", -1), line["fps_sync_every_frame"], line["fps_single_src"], line.get("memory_bytes")))
PY
done
