#!/bin/bash
mkdir -p gpurun_out/c11
export PYTHONPATH=.
O=gpurun_out/c11
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c11/bench_default.json').read().strip().splitlines()[-1])
print('headline ms', round(d['ms_per_step'], 2), 'value', round(d['value']), 'regions/s', round(d['regions_per_s']), 'frac', round(d['roofline']['frac'], 4), round(d['roofline']['frac_executed'], 4), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'], 'cpu', d['cpu_baseline'] and round(d['cpu_baseline']['value']))
for s in d.get('secondary', []):
    if 'error' in s:
        print(s); continue
    print(s['name'], 'ms', round(s['ms_per_step'], 1), 'regions/s', round(s['regions_per_s']), 'LP/s', round(s['value']), 'frac', round(s['roofline']['frac'], 4), 'cpu', s['cpu_baseline'] and round(s['cpu_baseline']['value']), 'wall', round(s['wall_seconds'], 1))
PY
