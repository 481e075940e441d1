cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.txt 2>&1
echo pytest_exit=$?
tail -40 gpurun_out/pytest_gpu.txt
