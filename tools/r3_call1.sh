#!/bin/bash
# round 3, first device call: midpoint table validation + config-5 status at the stated tolerance
mkdir -p gpurun_out/c1
export PYTHONPATH=.
O=gpurun_out/c1
timeout 300 python -m pytest tests/test_gpu_kernel_generations.py -k shared_midpoint -x -q -s > $O/midtable_test.txt 2>&1
echo "midtable test rc=$?" | tee -a $O/midtable_test.txt
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_table.json 2> $O/bench_table.err
echo "bench(table) rc=$?"
EHM_NO_MIDTABLE=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_notable.json 2> $O/bench_notable.err
echo "bench(no table) rc=$?"
python - <<'PY'
import json
for f in ('bench_table', 'bench_notable'):
    try:
        d = json.loads(open('gpurun_out/c1/%s.json' % f).read().strip().splitlines()[-1])
        c = d['config']
        print(f, 'ms', round(d['ms_per_step'], 2), 'LPs', c['lp_solves_per_step'], 'shared', c.get('midpoint_optima_taken_from_the_table_per_step'), 'nodes', c['nodes_per_step'], 'regions', c['regions_per_step'], 'frac', round(d['roofline']['frac'], 4), round(d['roofline']['frac_executed'], 4))
    except Exception as e:
        print(f, 'unreadable', e)
PY
timeout 400 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_partition.py -x -q > $O/parity.txt 2>&1
echo "parity rc=$?"; tail -2 $O/parity.txt
timeout 200 python tools/_seq_probe.py 150000 frontier > $O/seq_probe_loose.txt 2>&1
echo "seq probe rc=$?"; tail -1 $O/seq_probe_loose.txt | cut -c1-600
timeout 300 python tools/config5_probe.py 0.1 1e-3 3000 > $O/config5_tight.txt 2>&1
echo "config5 tight rc=$?"; tail -25 $O/config5_tight.txt
