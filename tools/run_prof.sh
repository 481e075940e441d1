# rocprofv3 kernel-trace + stats of the default bench command (run on the GPU box)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hd -- python3 $R/bench.py --steps 100 --warmup 10 --mode ${MODE:-graph} --no-cpu-baseline > $R/gpurun_out/prof_hd_stdout.txt 2>&1
echo exit=$?
ls -R $R/gpurun_out/prof_hd | head -30
