# The two ways bench.py is started with N ranks (torch.distributed.run, and bare --gpus N), rehearsed with 2 ranks on ONE GPU over gloo.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
RVM_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/mp2.txt 2>&1 || { tail -30 gpurun_out/mp2.txt; exit 1; }
tail -n 1 gpurun_out/mp2.txt | cut -c1-700
RVM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/mp2b.txt 2>&1 || { tail -30 gpurun_out/mp2b.txt; exit 1; }
tail -n 1 gpurun_out/mp2b.txt | cut -c1-300
