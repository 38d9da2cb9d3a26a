cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# two ranks sharing the one GPU of the box (gloo carries the collectives): exercises the
# N > 1 path of bench.py end to end -- dealing, rebalancing rounds, counter reductions
EHM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --abs-frac 0.05 ${BENCH_EXTRA:-} 2>&1 | tail -3 | cut -c1-1800
