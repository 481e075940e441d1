# Frame pipeline with four encoder stages: what the stream probe finds and the rate, with 4 and 8 hardware queues per process
# and with / without the stage graphs.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" RVM_VERBOSE=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-also --steps 300 --pipeline on > gpurun_out/pipe4.txt 2>&1 || { tail -30 gpurun_out/pipe4.txt; exit 1; }
  grep "frame pipeline" gpurun_out/pipe4.txt | sort | uniq -c
  python - <<'PY'
import json
line = json.loads(open("gpurun_out/pipe4.txt").read().strip().splitlines()[-1])
print("   %.1f frames/s (%.1f us/step)" % (line["value"], line["ms_per_step"] * 1e3))
PY
}
run GPU_MAX_HW_QUEUES=8 RVM_PIPE_STAGES=4
run GPU_MAX_HW_QUEUES=8 RVM_PIPE_STAGES=4 RVM_PIPE_GRAPH=0
run GPU_MAX_HW_QUEUES=8 RVM_PIPE_STAGES=4 RVM_PIPE_PROBE=0
run GPU_MAX_HW_QUEUES=4 RVM_PIPE_STAGES=4
run GPU_MAX_HW_QUEUES=8 RVM_PIPE_STAGES=3
