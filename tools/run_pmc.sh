# PMC counter passes over one bench workload (one pass per run, counters only - no tracing)
#   WL=hd|4k|hd_resnet50  TAG=name  BATCH=n  CHUNK=n (streams per call / frames per chunk, default 1)
# RVM_AUTOTUNE=0: model-chosen tiles, so that the three passes (separate processes) run the very same kernel sequence
export RVM_AUTOTUNE=0
R=$GRAFT_REPO_ROOT
TAG=${TAG:-hd}
mkdir -p $R/gpurun_out/pmc_$TAG
export TMPDIR=/tmp
cd /tmp
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/bench.py --steps 12 --warmup 4 --pipeline off --one-schedule --mode eager --workload ${WL:-hd} --batch ${BATCH:-1} --chunk ${CHUNK:-1} --no-cpu-baseline --no-also --dump-kernels $R/gpurun_out/pmc_$TAG/kernels.json > $R/gpurun_out/pmc_$TAG/stdout_$i.txt 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_$TAG/stdout_$i.txt; exit 1; }
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmc_$TAG gpurun_out/pmc_$TAG/kernels.json > gpurun_out/pmc_${TAG}_table.txt 2>&1
python3 tools/pmc_table.py gpurun_out/pmc_$TAG gpurun_out/pmc_$TAG/kernels.json --json > gpurun_out/pmc_${TAG}.json 2>/dev/null
rm -rf gpurun_out/pmc_$TAG/p1 gpurun_out/pmc_$TAG/p2 gpurun_out/pmc_$TAG/p3
head -6 gpurun_out/pmc_${TAG}_table.txt
