cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --abs-frac 0.05 2>&1 | tail -2 | cut -c1-600
EHM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --abs-frac 0.05 2>&1 | tail -3 | cut -c1-1500
