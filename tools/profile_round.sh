#!/bin/bash
# The profile pass of a round (gpurun -- 'bash tools/profile_round.sh'): kernel statistics + PMC
# counters of the headline, config 3, config 4 and the quadratic workload (tools/profile.sh), the
# static 8-way deal emulated on one GPU (headline tree, the 6.9 M-node tree, config 4), and the
# N > 1 path of bench.py end to end with two gloo ranks sharing the GPU.  Everything lands under
# gpurun_out/; copy what is to be judged into profiles/<round>/ (profiles/README.md says which
# file is which).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
O=$R/gpurun_out/round
mkdir -p $O
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.err
RND=${RND:-r6}
BENCH_ARGS="" bash $R/tools/profile.sh ${RND}_bench
BENCH_ARGS="--workload config4" bash $R/tools/profile.sh ${RND}_wide
BENCH_ARGS="--workload config3" bash $R/tools/profile.sh ${RND}_config3
BENCH_ARGS="--workload config2q" bash $R/tools/profile.sh ${RND}_quad
BENCH_ARGS="--workload config5" bash $R/tools/profile.sh ${RND}_config5
BENCH_ARGS="--workload explicit --queries 1048576" PROFILE_ARGS="--no-cpu-baseline" bash $R/tools/profile.sh ${RND}_explicit
cd $R
# the LDS-resident wide solver: phase table of an -DEHM4_PROFILE build (tools/k4_phases.py build,
# done on the build host: the library travels with the snapshot), the two families side by side,
# and the pipe micro-benchmarks its design is priced with
if [ -f $R/explicit_hybrid_mpc_amd/lib/libehmpc_k4prof.so ]; then
  EHM_LIB=$R/explicit_hybrid_mpc_amd/lib/libehmpc_k4prof.so timeout 300 python tools/k4_phases.py chain > $O/k4_phases_chain.json 2> $O/k4_phases.err
fi
timeout 400 python tools/k4_check.py chain > $O/k4_check_chain.json 2> $O/k4_check.err
[ -x $R/tools/micro/pipe_rates.bin ] && timeout 60 $R/tools/micro/pipe_rates.bin > $O/pipe_rates.txt 2>&1
timeout 600 python bench.py --workload explicit --steps 3 --cpu-seconds 8 > $O/bench_explicit.json 2> $O/bench_explicit.err
timeout 200 python tools/shard_balance.py deal > $O/shard_balance_deal_1p6M_nodes.txt 2>&1
tail -4 $O/shard_balance_deal_1p6M_nodes.txt
timeout 300 python tools/shard_balance.py deal 0.012 > $O/shard_balance_deal_6p9M_nodes.txt 2>&1
tail -4 $O/shard_balance_deal_6p9M_nodes.txt
timeout 300 python tools/shard_balance.py sweeps 0.4 config4 > $O/shard_balance_config4.txt 2>&1
tail -7 $O/shard_balance_config4.txt
for bal in static dynamic; do
  EHM_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 \
      --balance $bal > $O/bench_2ranks_$bal.json 2> $O/bench_2ranks_$bal.err
  echo "2 gloo ranks on one GPU ($bal): rc=$?"
done
