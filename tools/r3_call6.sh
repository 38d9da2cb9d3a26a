#!/bin/bash
mkdir -p gpurun_out/c6
export PYTHONPATH=.
O=gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_kernel_generations.py "tests/test_gpu_sequences.py::test_whole_cell_partition_delivers_the_guarantee" tests/test_gpu_hybrid.py::test_config3_subforests_identical_to_cpu_oracle -x -q -s > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -v "^$" $O/tests.txt | tail -25 | cut -c1-400
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt; tail -5 $O/bench_full.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c6/bench_full.json').read().strip().splitlines()[-1])
print('headline ms', round(d['ms_per_step'], 2), 'value', d['value'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
for s in d.get('secondary', []):
    if 'error' in s:
        print(s); continue
    print(s['name'], 'ms', round(s['ms_per_step'], 1), 'regions/s', round(s['regions_per_s']), 'LP/s', round(s['value']), 'frac', round(s['roofline']['frac'], 4), 'cpu', s['cpu_baseline'] and round(s['cpu_baseline']['value']), 'wall', round(s['wall_seconds'], 1), {k: s['config'].get(k) for k in ('regions_per_step', 'open_leaves_at_max_depth_per_step', 'lp_solves_per_mixed_integer_oracle_call')})
PY
