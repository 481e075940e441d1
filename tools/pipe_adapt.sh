# Frame pipeline: frames that find the device idle go to the caller's stream (RVM_PIPE_ADAPT, default on) - rates with and without,
# for callers that never wait and callers that wait for every frame.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "hd --steps 300" "hd --steps 300 --dtype fp16fast" "hd --batch 8 --steps 100" "4k --steps 200" "hd_resnet50 --steps 100"; do
set -- $cfg; wl=$1; shift
for ad in 1 0; do
RVM_PIPE_ADAPT=$ad timeout -k 10 300 python -u bench.py --workload $wl --no-cpu-baseline --no-also "$@" > gpurun_out/pipe_adapt.txt 2>&1 || { tail -30 gpurun_out/pipe_adapt.txt; exit 1; }
python - "$cfg" $ad <<'PY'
import json, sys
line = json.loads(open("gpurun_out/pipe_adapt.txt").read().strip().splitlines()[-1])
print("%-36s adapt=%s: %.1f frames/s   one stream %.1f   sync every frame: pipelined %.1f, one stream %.1f" % (sys.argv[1], sys.argv[2], line["value"], line["fps_pipeline_off"], line["fps_sync_every_frame_pipelined"], line["fps_sync_every_frame"]))
PY
done; done
