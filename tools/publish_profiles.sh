# Copy the outputs of tools/collect_profiles.sh (gpurun_out/final_*, prof_*, pmc_*) into profiles/<round>/.
#   bash tools/publish_profiles.sh r03
R=${1:-r03}; P=profiles/$R; mkdir -p $P
for w in hd 4k rn hd_b8 hd_c12; do
  [ -f gpurun_out/prof_${w}_table.txt ] && cp gpurun_out/prof_${w}_table.txt $P/${w}_kernel_trace_table.txt
  [ -f gpurun_out/prof_${w}_kernel_stats.csv ] && cp gpurun_out/prof_${w}_kernel_stats.csv $P/${w}_kernel_stats.csv
done
for w in hd 4k rn hd_b8 hd_c12; do
  [ -f gpurun_out/pmc_$w.json ] && cp gpurun_out/pmc_$w.json $P/pmc_$w.json && cp gpurun_out/pmc_${w}_table.txt $P/pmc_${w}_table.txt
  [ -f gpurun_out/final_stamps_$w.txt ] && grep -v "^/opt" gpurun_out/final_stamps_$w.txt > $P/${w}_inkernel_timeline.txt
  [ -f gpurun_out/final_kernels_$w.json ] && cp gpurun_out/final_kernels_$w.json $P/${w}_per_launch_events.json
done
for n in hd 4k hd_fast hd_bf16 hd_fp32 hd_b8 hd_chunk12 4k_chunk8 rn hd_eager hd_one_stream driver_4k; do
  [ -f gpurun_out/final_bench_$n.txt ] && tail -n 1 gpurun_out/final_bench_$n.txt > $P/bench_$n.json
done
ls $P
