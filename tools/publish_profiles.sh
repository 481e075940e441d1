# Copy the outputs of tools/collect_profiles.sh (gpurun_out/final_*, prof_*, pmc_*) into profiles/<round>/ as version <tag>.
#   bash tools/publish_profiles.sh r01 v3
R=${1:-r01}; V=${2:-v3}; P=profiles/$R; mkdir -p $P
cp gpurun_out/prof_hd_table.txt $P/hd_kernel_trace_table_$V.txt
cp gpurun_out/prof_4k_table.txt $P/4k_kernel_trace_table_$V.txt
cp gpurun_out/prof_hd_kernel_stats.csv $P/hd_kernel_stats_$V.csv
cp gpurun_out/prof_4k_kernel_stats.csv $P/4k_kernel_stats_$V.csv
cp gpurun_out/pmc_hd.json $P/pmc_hd_$V.json; cp gpurun_out/pmc_4k.json $P/pmc_4k_$V.json
cp gpurun_out/pmc_hd_table.txt $P/pmc_hd_${V}_table.txt; cp gpurun_out/pmc_4k_table.txt $P/pmc_4k_${V}_table.txt
grep -v "^/opt" gpurun_out/final_stamps_hd.txt > $P/hd_inkernel_timeline_$V.txt
grep -v "^/opt" gpurun_out/final_stamps_4k.txt > $P/4k_inkernel_timeline_$V.txt
for n in hd 4k hd_fast hd_fp32 hd_b8 hd_chunk12 4k_chunk8 rn hd_eager; do tail -n 1 gpurun_out/final_bench_$n.txt > $P/bench_${n}_$V.json; done
cp gpurun_out/final_kernels_hd.json $P/hd_per_launch_events_$V.json; cp gpurun_out/final_kernels_4k.json $P/4k_per_launch_events_$V.json
