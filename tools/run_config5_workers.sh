set -x
export EHM_BENCH_BACKEND=gloo
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --workload config5 --cells 4 --cells-at-once 1 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/c5_w2_c4.json 2> gpurun_out/c5_w2_c4.err
rc=$?
tail -3 gpurun_out/c5_w2_c4.err
python -c "
import json; d=json.load(open('gpurun_out/c5_w2_c4.json')); print('SMALL', d['ms_per_step'], d['regions_per_s'], d['config']['regions_per_step'], d['roofline']['device_share_of_the_step'], d['n_gpus'], d['host_processes'])" || exit 1
timeout 460 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 4 --workload config5 --cells 40 --cells-at-once 1 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_config5_1e6_regions.json 2> gpurun_out/bench_config5_1e6.err
tail -3 gpurun_out/bench_config5_1e6.err
python -c "
import json; d=json.load(open('gpurun_out/bench_config5_1e6_regions.json')); print('FULL', d['ms_per_step'], d['regions_per_s'], d['config']['regions_per_step'], d['config']['open_leaves_per_step'], d['roofline']['device_share_of_the_step'], d['roofline']['tables'])"
