cd $GRAFT_REPO_ROOT
for S in 2 1; do
python bench.py --workload config2q --solver $S --abs-frac 0.1 --eps-r 0.1 --steps 3 --warmup 1 --cpu-seconds 10 $( [ $S = 1 ] && echo --no-cpu-baseline ) > gpurun_out/bench_config2q_s$S.json 2> gpurun_out/bench_config2q_s$S.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_config2q_s$S.json'))
print('solver $S', d['value'], d['regions_per_s'], d['ms_per_step'], d['config']['mean_ipm_iterations'], d['roofline']['frac'], d['config']['kernels'], d['config']['suboptimality_test'])
PY
done
(time python -m pytest tests -m gpu -x -q) > gpurun_out/gputests.log 2>&1; tail -4 gpurun_out/gputests.log
