cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export RVM_AUTOTUNE=0
for b in 0 1; do
RVM_STAGE_BALANCE=$b timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/q_stamps_hd_b$b.txt 2>&1 || { tail -30 gpurun_out/q_stamps_hd_b$b.txt; exit 1; }
echo "balance $b"; grep -E "frame |decode[0-9].conv|gru" gpurun_out/q_stamps_hd_b$b.txt | cut -c1-110
done
RVM_STAGE_BALANCE=1 timeout -k 10 200 python -u tools/gpu_check.py --dtype fp32 --H 150 --W 201 --ds 0.8 --T 2 --B 2 > gpurun_out/q_bal.txt 2>&1 || { tail gpurun_out/q_bal.txt; exit 1; }
tail -1 gpurun_out/q_bal.txt | cut -c 150-400
