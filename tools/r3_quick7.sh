cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RVM_AUTOTUNE=0 RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd_model.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd_model.txt; exit 1; }
grep -E "frame |decode[0-9].conv|conv_cat" gpurun_out/q_stamps_hd_model.txt | sort -u | cut -c1-150
RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd.txt; exit 1; }
grep -E "frame |decode[0-9].conv |conv_cat" gpurun_out/q_stamps_hd.txt | grep -v "rvm tune" | sort -u | cut -c1-150
RVM_AUTOTUNE=0 RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py --workload 4k > gpurun_out/q_stamps_4k_model.txt 2>&1 || { tail -30 gpurun_out/q_stamps_4k_model.txt; exit 1; }
grep -E "frame |decode[0-9].conv|conv_cat" gpurun_out/q_stamps_4k_model.txt | sort -u | cut -c1-150
RVM_VERBOSE=1 timeout -k 10 200 python -u tools/stamp_table.py --workload 4k > gpurun_out/q_stamps_4k.txt 2>&1 || { tail -30 gpurun_out/q_stamps_4k.txt; exit 1; }
grep -E "frame |decode[0-9].conv |conv_cat" gpurun_out/q_stamps_4k.txt | grep -v "rvm tune" | sort -u | cut -c1-150
