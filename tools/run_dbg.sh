# ablation timing: RVM_DBG bit sets (conv_kxk: bits 0-2, mbconv_ed: bits 4-6), rocprof kernel trace per setting
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/dbg
export TMPDIR=/tmp
cd /tmp
for d in ${DBGS:-0 16 32 64 112}; do
  RVM_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dbg/p_$d -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --dump-kernels $R/gpurun_out/dbg/k_$d.json > $R/gpurun_out/dbg/b_$d.txt 2>&1 || exit 1
  f=$(ls $R/gpurun_out/dbg/p_$d/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_table.py $f $R/gpurun_out/dbg/k_$d.json 30 > $R/gpurun_out/dbg/t_$d.txt 2>&1
  rm -rf $R/gpurun_out/dbg/p_$d
done
echo ok
