# Frame pipeline: events against flag kernels (RVM_PIPE_SYNC=kernel) for ordering the stage streams; 3 fresh processes per point.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--steps 300" "--steps 20 --warmup 5" "--workload 4k --steps 200" "--batch 8 --steps 100" "--workload hd_resnet50 --steps 100"; do for sync in event kernel; do
out=""
for rep in 1 2 3; do
RVM_PIPE_SYNC=$sync timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-also $args > gpurun_out/syncm.txt 2>&1 || { tail -30 gpurun_out/syncm.txt; exit 1; }
out="$out $(python -c "import json;l=json.loads(open('gpurun_out/syncm.txt').read().strip().splitlines()[-1]);print('%.0f/%.0f' % (l['value'], l['fps_sync_every_frame_pipelined']))")"
done
echo "$args sync=$sync:$out"
done; done
