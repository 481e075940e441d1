# Frame pipeline: the pipelined HD rate against the number of streams the process made before (RVM_PIPE_DUMMY), with the
# stage streams picked by probing (default) and taken as they come (RVM_PIPE_PROBE=0).  Three fresh processes per point.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pr in ${PROBE:-1 0}; do for st in ${STAGES:-2 3}; do for d in 0 1 2 3 4; do
out=""
for rep in 1 2 3; do
RVM_VERBOSE=1 RVM_PIPE_PROBE=$pr RVM_PIPE_STAGES=$st RVM_PIPE_DUMMY=$d timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-also --steps 300 > gpurun_out/queues.txt 2>&1 || { tail -30 gpurun_out/queues.txt; exit 1; }
out="$out $(python -c "import json;l=json.loads(open('gpurun_out/queues.txt').read().strip().splitlines()[-1]);print('%.0f' % l['value'])")"
done
echo "probe=$pr stages=$st streams before=$d:$out   $(grep -h 'stage streams' gpurun_out/queues.txt | tail -1 | cut -c7-)"
done; done; done
