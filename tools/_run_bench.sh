cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# two ranks sharing the one GPU of the box (gloo carries the collectives): exercises the
# N > 1 paths of bench.py end to end -- static dealing + persistent kernel per rank, and dealing,
# rebalancing rounds, counter reductions of the dynamic mode
for B in static dynamic; do
EHM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --abs-frac 0.05 --balance $B ${BENCH_EXTRA:-} 2>gpurun_out/bench_n2_$B.err | grep '^{' | tail -1 > gpurun_out/bench_n2_$B.json
python - <<PY
import json
d=json.load(open('gpurun_out/bench_n2_$B.json'))
c=d['config']
print('$B', d['n_gpus'], d['value'], d['ms_per_step'], c['regions_per_step'], c['nodes_per_step'], c['lp_solves_per_step'], c['lp_solves_per_rank'], c['load_imbalance_max_over_mean'], c['parallelism'][:60])
PY
done
python bench.py --steps 2 --warmup 1 --abs-frac 0.05 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('single', d['value'], d['ms_per_step'], c['regions_per_step'], c['nodes_per_step'], c['lp_solves_per_step'])"
