#!/bin/bash
# profile.sh for the listed workloads + their un-profiled bench lines (gpurun -- 'bash tools/_profile_all.sh bench wide')
cd ${GRAFT_REPO_ROOT:-.}
for TAG in "$@"; do
  case $TAG in
    bench)   ARGS="";                     OUT=bench_n1 ;;
    wide)    ARGS="--workload config4";   OUT=bench_config4 ;;
    config3) ARGS="--workload config3";   OUT=bench_config3 ;;
    quad)    ARGS="--workload config2q";  OUT=bench_config2q ;;
  esac
  BENCH_ARGS="$ARGS" bash tools/profile.sh $TAG
  mkdir -p profiles/r2
  cp gpurun_out/prof_${TAG}_summary.json profiles/r2/pmc_summary_${TAG}.json
  if [ "$TAG" = bench ]; then
    python bench.py $ARGS > gpurun_out/${OUT}.json 2> gpurun_out/${OUT}.err
  else
    python bench.py --no-cpu-baseline $ARGS > gpurun_out/${OUT}.json 2> gpurun_out/${OUT}.err
  fi
  tail -c 600 gpurun_out/${OUT}.json
  find gpurun_out/prof_$TAG/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${OUT}_kernel_stats.csv
done
