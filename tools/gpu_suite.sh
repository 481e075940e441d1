# Full GPU pass (GPU box): every -m gpu test, the allocation-free entry point's test alone in a fresh process.
#   PART=a: all tests but the 300-frame 4K drift test   PART=b: that test alone (timed)   default: a
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ "${PART:-a}" = "a" ]; then
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -v --timeout 400 --timeout-method thread --durations=15 -k "not test_config2_4k_fp16_T300_drift" > gpurun_out/r3_pytest.txt 2>&1 || { tail -40 gpurun_out/r3_pytest.txt; exit 1; }
tail -22 gpurun_out/r3_pytest.txt
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -x -v --timeout 100 --timeout-method thread > gpurun_out/r3_noalloc.txt 2>&1 || { tail -40 gpurun_out/r3_noalloc.txt; exit 1; }
tail -4 gpurun_out/r3_noalloc.txt
else
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 800 --timeout-method thread --durations=3 -k "test_config2_4k_fp16_T300_drift" > gpurun_out/r3_pytest_4k300.txt 2>&1 || { tail -40 gpurun_out/r3_pytest_4k300.txt; exit 1; }
tail -8 gpurun_out/r3_pytest_4k300.txt
fi
