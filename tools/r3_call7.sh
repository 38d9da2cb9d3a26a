#!/bin/bash
mkdir -p gpurun_out/c7
export PYTHONPATH=.
O=gpurun_out/c7
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c7/bench.json').read().strip().splitlines()[-1]); c = d['config']
print('ms', round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 4), 'LPs', c['lp_solves_per_step'], 'nodes', c['nodes_per_step'], 'regions', c['regions_per_step'])
print(d.get('persist_ticks'))
PY
timeout 1200 python -m pytest tests/test_gpu_kernel_generations.py tests/test_gpu_partition.py tests/test_gpu_bench_parity.py tests/test_gpu_quadratic.py "tests/test_gpu_sequences.py::test_whole_cell_partition_delivers_the_guarantee" tests/test_gpu_hybrid.py::test_config3_subforests_identical_to_cpu_oracle -q -s > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -v "^$" $O/tests.txt | grep "passed\|failed\|FAILED\|N=8\|guarantee\|visited again\|share_midpoints" | cut -c1-600
timeout 600 python bench.py --workload config5 --cells 4 --steps 1 --warmup 0 --cpu-seconds 10 > $O/bench_config5.json 2> $O/bench_config5.err; echo "config5 rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/c7/bench_config5.json').read().strip().splitlines()[-1]); c = d['config']
    print('config5 ms', round(d['ms_per_step']), 'regions', c['regions_per_step'], 'nodes', c['nodes_per_step'], 'LPs', c['lp_solves_per_step'], 'LP/MICP', round(c['lp_solves_per_mixed_integer_oracle_call'], 1), 'frac', round(d['roofline']['frac'], 4), 'device share', round(d['roofline']['device_share_of_the_step'], 3), 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], c['mixed_integer_oracle_calls_per_step'])
except Exception as e:
    print('config5 unreadable', e)
PY
tail -3 $O/bench_config5.err
