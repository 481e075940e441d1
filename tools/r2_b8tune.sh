cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RVM_VERBOSE=1 timeout -k 10 300 python -u bench.py --batch 8 --steps 50 --no-also --no-cpu-baseline > gpurun_out/b8_tune.txt 2>&1 || { tail -20 gpurun_out/b8_tune.txt; exit 1; }
tail -n 1 gpurun_out/b8_tune.txt | cut -c1-300
