#!/bin/bash
# The profile pass of a round (gpurun -- 'bash tools/profile_round.sh'): kernel statistics + PMC
# counters of the headline, config 3 and config 4 (tools/profile.sh), kernel statistics of config 5
# (cell 0 to completion), the static 8-way deal emulated on one GPU, and the N > 1 path of bench.py
# end to end with two gloo ranks sharing the GPU.  Everything lands under gpurun_out/; copy what is
# to be judged into profiles/<round>/ (profiles/README.md says which file is which).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
O=$R/gpurun_out/round
mkdir -p $O
BENCH_ARGS="" bash $R/tools/profile.sh r3_bench
BENCH_ARGS="--workload config4" bash $R/tools/profile.sh r3_wide
BENCH_ARGS="--workload config3" bash $R/tools/profile.sh r3_config3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_config5/stats -- \
    python $R/bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline \
    > $O/bench_config5_cell0.json 2> $O/config5.err
cd $R
timeout 300 python tools/shard_balance.py deal > $O/shard_balance_deal.txt 2>&1
tail -13 $O/shard_balance_deal.txt
for bal in static dynamic; do
  EHM_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 \
      --balance $bal > $O/bench_2ranks_$bal.json 2> $O/bench_2ranks_$bal.err
  echo "2 gloo ranks on one GPU ($bal): rc=$?"
done
