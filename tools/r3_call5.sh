#!/bin/bash
mkdir -p gpurun_out/c5
export PYTHONPATH=.
O=gpurun_out/c5
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c5/bench.json').read().strip().splitlines()[-1]); c = d['config']
print('ms', round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 4), 'LPs', c['lp_solves_per_step'], c['lp_solves_by_kind_per_step'], 'table wit', c['nodes_proved_open_by_another_edges_midpoint_in_the_table_per_step'], 'inh', c['nodes_proved_open_by_inherited_witness_per_step'], 'mid', c['nodes_proved_open_by_midpoint_per_step'])
print(d.get('persist_ticks'))
PY
timeout 500 python -m pytest tests/test_gpu_kernel_generations.py tests/test_gpu_bench_parity.py tests/test_gpu_partition.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 420 python tools/config5_probe.py 0.15 1e-3 400000 256 2048 backoff lcss-first 120000 > $O/config5_a015.txt 2>&1
echo "config5 0.15 rc=$?"; tail -22 $O/config5_a015.txt
timeout 420 python tools/config5_probe.py 0.2 1e-3 400000 256 2048 backoff lcss-first 120000 > $O/config5_a02.txt 2>&1
echo "config5 0.2 rc=$?"; tail -22 $O/config5_a02.txt
