cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dbg
for d in 0 1 2 4 3 7; do
RVM_DBG=$d timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 100 --mode eager --dump-kernels gpurun_out/dbg/k_$d.json > gpurun_out/dbg/b_$d.txt 2>&1 || exit 1
done
echo ok
