cd $GRAFT_REPO_ROOT
python bench.py --workload config2q --abs-frac 0.1 --eps-r 0.1 --steps 2 --warmup 1 --cpu-seconds 10 > gpurun_out/bench_config2q.json 2> gpurun_out/bench_config2q.err; tail -c 2500 gpurun_out/bench_config2q.json; tail -3 gpurun_out/bench_config2q.err
