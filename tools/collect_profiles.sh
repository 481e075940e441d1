# Round-end evidence run (GPU box): kernel traces, PMC passes, in-kernel timelines and bench lines of the current build.
# Output under gpurun_out/; tools/publish_profiles.sh copies what is to be judged into profiles/<round>/.
#   PART=1: traces + timelines + bench lines   PART=2: PMC passes (hd, 4k, rn, hd_b8, hd_c12)   default: both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ "${PART:-1}" != "2" ]; then
TAG=hd bash tools/run_prof.sh > gpurun_out/final_prof_hd.log 2>&1 || exit 1
TAG=4k WL=4k bash tools/run_prof.sh > gpurun_out/final_prof_4k.log 2>&1 || exit 1
TAG=rn WL=hd_resnet50 bash tools/run_prof.sh > gpurun_out/final_prof_rn.log 2>&1 || exit 1
TAG=hd_b8 BATCH=8 bash tools/run_prof.sh > gpurun_out/final_prof_hd_b8.log 2>&1 || exit 1
TAG=hd_c12 CHUNK=12 bash tools/run_prof.sh > gpurun_out/final_prof_hd_c12.log 2>&1 || exit 1
timeout -k 10 200 python -u tools/stamp_table.py > gpurun_out/final_stamps_hd.txt 2>&1 || exit 1
timeout -k 10 200 python -u tools/stamp_table.py --workload 4k > gpurun_out/final_stamps_4k.txt 2>&1 || exit 1
timeout -k 10 200 python -u tools/stamp_table.py --workload hd_resnet50 > gpurun_out/final_stamps_rn.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --workload 4k --dump-kernels gpurun_out/final_kernels_4k.json > gpurun_out/final_bench_4k.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --dtype fp16fast > gpurun_out/final_bench_hd_fast.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --dtype bf16 > gpurun_out/final_bench_hd_bf16.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --dtype fp32 --steps 100 > gpurun_out/final_bench_hd_fp32.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --batch 8 --steps 100 --dump-kernels gpurun_out/final_kernels_hd_b8.json > gpurun_out/final_bench_hd_b8.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --chunk 12 --steps 50 > gpurun_out/final_bench_hd_chunk12.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --workload 4k --chunk 8 --steps 30 > gpurun_out/final_bench_4k_chunk8.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --workload hd_resnet50 --steps 100 --dump-kernels gpurun_out/final_kernels_rn.json > gpurun_out/final_bench_rn.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --mode eager > gpurun_out/final_bench_hd_eager.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-also --pipeline off --mode eager > gpurun_out/final_bench_hd_one_stream.txt 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --workload 4k --driver-pipeline > gpurun_out/final_bench_driver_4k.txt 2>&1 || exit 1
timeout -k 10 400 python -u bench.py --dump-kernels gpurun_out/final_kernels_hd.json > gpurun_out/final_bench_hd.txt 2>&1 || exit 1
for f in gpurun_out/final_bench_*.txt; do echo "$f $(tail -n 1 $f | cut -c80-150)"; done
fi
if [ "${PART:-2}" != "1" ]; then
TAG=hd bash tools/run_pmc.sh > gpurun_out/final_pmc_hd.log 2>&1 || exit 1
TAG=4k WL=4k bash tools/run_pmc.sh > gpurun_out/final_pmc_4k.log 2>&1 || exit 1
TAG=rn WL=hd_resnet50 bash tools/run_pmc.sh > gpurun_out/final_pmc_rn.log 2>&1 || exit 1
TAG=hd_b8 BATCH=8 bash tools/run_pmc.sh > gpurun_out/final_pmc_hd_b8.log 2>&1 || exit 1
TAG=hd_c12 CHUNK=12 bash tools/run_pmc.sh > gpurun_out/final_pmc_hd_c12.log 2>&1 || exit 1
head -3 gpurun_out/pmc_hd_table.txt gpurun_out/pmc_4k_table.txt gpurun_out/pmc_rn_table.txt
fi
