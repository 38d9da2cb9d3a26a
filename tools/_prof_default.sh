cd $GRAFT_REPO_ROOT
bash tools/profile.sh v3a
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 1800 gpurun_out/bench_default.json
