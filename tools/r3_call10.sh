#!/bin/bash
# round 3 profile pass: kernel statistics + PMC counters of the headline, config 3 and config 4;
# kernel statistics of config 5 (cell 0 to completion); static 8-way deal emulated on one GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
mkdir -p $R/gpurun_out/c10
BENCH_ARGS="" bash $R/tools/profile.sh r3_bench
BENCH_ARGS="--workload config4" bash $R/tools/profile.sh r3_wide
BENCH_ARGS="--workload config3" bash $R/tools/profile.sh r3_config3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_config5/stats -- python $R/bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/c10/bench_config5_cell0.json 2> $R/gpurun_out/c10/config5.err
cd $R
find gpurun_out/prof_r3_config5 -name "*kernel_stats.csv" | head -2
timeout 300 python tools/shard_balance.py deal > gpurun_out/c10/shard_balance_deal.txt 2>&1; tail -14 gpurun_out/c10/shard_balance_deal.txt
ls gpurun_out | head -30
