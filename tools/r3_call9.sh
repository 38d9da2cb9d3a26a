#!/bin/bash
mkdir -p gpurun_out/c9
export PYTHONPATH=.
O=gpurun_out/c9
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
    print(sys.argv[1], 'ms', round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 4), 'LPs', c['lp_solves_per_step'], 'nodes', c['nodes_per_step'], 'regions', c['regions_per_step'])
    t = d.get('persist_ticks') or {}
    print('   ', {k: round(v, 3) for k, v in t.items()})
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
}
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_wf.json 2> $O/bench_wf.err; show $O/bench_wf.json
EHM_NO_WORKFIRST=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_nowf.json 2> $O/bench_nowf.err; show $O/bench_nowf.json
timeout 200 python bench.py --workload config2q --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q.json 2> $O/bench_q.err; show $O/bench_q.json
timeout 900 python -m pytest tests/test_gpu_kernel_generations.py tests/test_gpu_partition.py tests/test_gpu_bench_parity.py tests/test_gpu_rebalance.py tests/test_gpu_edge_cases.py -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 500 python bench.py --workload config5 --cells 4 --regions 100000 --order fifo --steps 1 --warmup 0 --cpu-seconds 10 > $O/bench_config5_1e5.json 2> $O/bench_config5_1e5.err; echo "config5 rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/c9/bench_config5_1e5.json').read().strip().splitlines()[-1]); c = d['config']
    print('config5 ms', round(d['ms_per_step']), 'regions', c['regions_per_step'], 'nodes', c['nodes_per_step'], 'LPs', c['lp_solves_per_step'], 'LP/MICP', round(c['lp_solves_per_mixed_integer_oracle_call'], 1), 'frac', round(d['roofline']['frac'], 4), 'device share', round(d['roofline']['device_share_of_the_step'], 3), 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], c['mixed_integer_oracle_calls_per_step'])
except Exception as e:
    print('config5 unreadable', e)
PY
tail -3 $O/bench_config5_1e5.err
