#!/bin/bash
mkdir -p gpurun_out/c8
export PYTHONPATH=.
O=gpurun_out/c8
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
    print(sys.argv[1], 'ms', round(d['ms_per_step'], 1), 'nodes', c['nodes_per_step'], 'regions', c['regions_per_step'], 'LPs', c['lp_solves_per_step'], 'shared', c.get('midpoint_optima_taken_from_the_table_per_step'), 'frac', round(d['roofline']['frac'], 4))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout 200 python bench.py --workload config3 --steps 3 --warmup 1 --no-cpu-baseline > $O/config3_table.json 2> $O/config3_table.err; show $O/config3_table.json
EHM_NO_MIDTABLE=1 timeout 200 python bench.py --workload config3 --steps 3 --warmup 1 --no-cpu-baseline > $O/config3_notable.json 2> $O/config3_notable.err; show $O/config3_notable.json
timeout 300 python tools/cwh_jobs.py 5 > $O/cwh_jobs_table.jsonl 2>&1; cut -c1-260 $O/cwh_jobs_table.jsonl
EHM_NO_MIDTABLE=1 timeout 300 python tools/cwh_jobs.py 4 > $O/cwh_jobs_notable.jsonl 2>&1; cut -c1-260 $O/cwh_jobs_notable.jsonl
timeout 1200 python -m pytest tests/test_gpu_kernel_generations.py tests/test_gpu_hybrid.py tests/test_gpu_quadratic.py tests/test_gpu_rebalance.py tests/test_gpu_api.py "tests/test_gpu_sequences.py::test_whole_cell_partition_delivers_the_guarantee" -q -s > $O/tests.txt 2>&1; echo "tests rc=$?"; grep "passed\|failed\|FAILED\|visited again\|^E " $O/tests.txt | cut -c1-400
for c in 1 2 3; do
  EHM_CELL=$c timeout 120 python tools/config5_probe.py 0.2 1e-3 150000 256 2048 backoff lcss-first > $O/config5_cell$c.txt 2>&1
  echo "cell $c rc=$?"; grep "^visits\|^calls\|   round" $O/config5_cell$c.txt | tail -4 | cut -c1-300
done
