# Plan-time measurements of one workload (GPU box): every candidate of every tuned launch, summarised by tools/tune_table.py.
#   WL=hd_resnet50 bash tools/tune_dump.sh            BATCH=8 bash tools/tune_dump.sh            FORCE=1: also for a single mobilenetv3 frame
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${WL:-hd}_b${BATCH:-1}
[ "${FORCE:-0}" = "1" ] && export RVM_AUTOTUNE=1
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --workload ${WL:-hd} --batch ${BATCH:-1} --steps 50 --no-also --no-cpu-baseline > gpurun_out/tune_$T.txt 2>&1 || { tail -20 gpurun_out/tune_$T.txt; exit 1; }
tail -n 1 gpurun_out/tune_$T.txt | cut -c1-200
python3 tools/tune_table.py gpurun_out/tune_$T.txt > gpurun_out/tune_${T}_table.txt
wc -l gpurun_out/tune_${T}_table.txt
