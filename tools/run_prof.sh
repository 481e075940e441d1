# rocprofv3 kernel-trace + stats of one bench command (run on the GPU box)
#   MODE=eager|graph  WL=hd|4k|hd_resnet50  DT=fp16|fp16fast|fp32|bf16  BATCH  CHUNK  TAG=name
R=$GRAFT_REPO_ROOT
TAG=${TAG:-hd}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 100 --warmup 10 --pipeline off --one-schedule --mode ${MODE:-eager} --workload ${WL:-hd} --dtype ${DT:-fp16} --batch ${BATCH:-1} --chunk ${CHUNK:-1} --no-cpu-baseline --no-also --dump-kernels $R/gpurun_out/prof_${TAG}_kernels.json > $R/gpurun_out/prof_${TAG}_stdout.txt 2>&1
echo exit=$?
cd $R
f=$(ls gpurun_out/prof_$TAG/*/*kernel_trace.csv | head -1)
python3 tools/trace_table.py $f gpurun_out/prof_${TAG}_kernels.json 50 > gpurun_out/prof_${TAG}_table.txt 2>&1
cp $(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/prof_${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
head -3 gpurun_out/prof_${TAG}_table.txt
