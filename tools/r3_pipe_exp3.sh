cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_frame_pipeline.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
for st in ${STAGES:-1 2 3}; do for sync in ${SYNC:-event kernel}; do
RVM_PIPE_STAGES=$st RVM_PIPE_SYNC=$sync timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-also --steps 300 --pipeline on ${ARGS} > gpurun_out/pipe_exp.txt 2>&1 || { tail -30 gpurun_out/pipe_exp.txt; exit 1; }
python - $st $sync <<'PY'
import json, sys
line = json.loads(open("gpurun_out/pipe_exp.txt").read().strip().splitlines()[-1])
print("stages=%s sync=%s: %.1f frames/s (%.1f us/step)  sync-every-frame %.1f us" % (sys.argv[1], sys.argv[2], line["value"], line["ms_per_step"] * 1e3, 1e6 / line["fps_sync_every_frame"]))
PY
done; done
