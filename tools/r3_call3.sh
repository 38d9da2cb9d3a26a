#!/bin/bash
mkdir -p gpurun_out/c3
export PYTHONPATH=.
O=gpurun_out/c3
timeout 400 python tools/config5_probe.py 0.1 1e-3 8000 256 1024 - deepest > $O/config5_deep.txt 2>&1
echo "config5 deepest rc=$?"; tail -16 $O/config5_deep.txt
timeout 400 python tools/config5_probe.py 0.1 1e-3 8000 256 1024 backoff deepest > $O/config5_deep_backoff.txt 2>&1
echo "config5 deepest backoff rc=$?"; tail -16 $O/config5_deep_backoff.txt
for c in "0.5 0.1 24" "0.25 0.1 24" "0.25 0.05 22" "0.1 0.01 18"; do
  set -- $c
  timeout 150 python bench.py --workload config3 --abs-frac $1 --eps-r $2 --max-depth $3 --steps 1 --warmup 0 --no-cpu-baseline > $O/c3_$1_$2_$3.json 2> $O/c3_$1_$2_$3.err
  python - "$O/c3_$1_$2_$3.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
    print(sys.argv[1], 'ms', round(d['ms_per_step'], 1), 'nodes', c['nodes_per_step'], 'regions', c['regions_per_step'], 'open', c['open_leaves_at_max_depth_per_step'], 'LPs', c['lp_solves_per_step'], 'depth', c['tree_depth'])
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
