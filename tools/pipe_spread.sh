# Frame pipeline: run-to-run spread of the pipelined rate (fresh process each), stages 2 / 3, fp16 / bf16.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for st in 2 3; do for dt in fp16 bf16; do
out=""
for rep in 1 2 3 4; do
RVM_PIPE_STAGES=$st timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-also --steps 300 --dtype $dt ${ARGS} > gpurun_out/pipe_stab.txt 2>&1 || { tail -30 gpurun_out/pipe_stab.txt; exit 1; }
out="$out $(python -c "import json;l=json.loads(open('gpurun_out/pipe_stab.txt').read().strip().splitlines()[-1]);print('%.0f' % l['value'])")"
done
echo "stages=$st $dt:$out"
done; done
