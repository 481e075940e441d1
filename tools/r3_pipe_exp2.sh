cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for sync in kernel event memop; do for only in 0 1 2; do
RVM_PIPE_SYNC=$sync RVM_PIPE_ONLY=$only timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-also --steps 300 --pipeline on ${ARGS} > gpurun_out/pipe_exp.txt 2>&1 || { tail -30 gpurun_out/pipe_exp.txt; exit 1; }
python - $sync $only <<'PY'
import json, sys
line = json.loads(open("gpurun_out/pipe_exp.txt").read().strip().splitlines()[-1])
print("sync=%s only=%s: %.1f frames/s (%.1f us/step)  sync-every-frame %.1f us" % (sys.argv[1], sys.argv[2], line["value"], line["ms_per_step"] * 1e3, 1e6 / line["fps_sync_every_frame"]))
PY
done; done
