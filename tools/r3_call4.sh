#!/bin/bash
mkdir -p gpurun_out/c4
export PYTHONPATH=.
O=gpurun_out/c4
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
    print(sys.argv[1], 'ms', round(d['ms_per_step'], 1), 'nodes', c['nodes_per_step'], 'regions', c['regions_per_step'], 'LPs', c['lp_solves_per_step'], 'shared', c.get('midpoint_optima_taken_from_the_table_per_step'), 'frac', round(d['roofline']['frac'], 4))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout 200 python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu-baseline > $O/config4_table.json 2> $O/config4_table.err; show $O/config4_table.json
EHM_NO_MIDTABLE=1 timeout 200 python bench.py --workload config4 --steps 3 --warmup 1 --no-cpu-baseline > $O/config4_notable.json 2> $O/config4_notable.err; show $O/config4_notable.json
timeout 300 python -m pytest tests/test_gpu_wide.py -x -q > $O/wide_tests.txt 2>&1; echo "wide tests rc=$?"; tail -2 $O/wide_tests.txt
timeout 300 python tools/config5_probe.py 0.25 1e-3 40000 256 2048 backoff lcss-first 20000 > $O/config5_a025.txt 2>&1
echo "config5 0.25 rc=$?"; tail -24 $O/config5_a025.txt
timeout 300 python tools/config5_probe.py 0.5 1e-3 40000 256 2048 backoff lcss-first 20000 > $O/config5_a05.txt 2>&1
echo "config5 0.5 rc=$?"; tail -24 $O/config5_a05.txt
