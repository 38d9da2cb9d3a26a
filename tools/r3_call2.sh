#!/bin/bash
mkdir -p gpurun_out/c2
export PYTHONPATH=.
O=gpurun_out/c2
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c2/bench.json').read().strip().splitlines()[-1])
print('ms', round(d['ms_per_step'], 2), 'frac', round(d['roofline']['frac'], 4), d.get('persist_ticks'))
PY
timeout 300 python tools/config5_probe.py 0.1 1e-3 6000 256 1024 - deepest > $O/config5_deep.txt 2>&1
echo "config5 deepest rc=$?"; tail -22 $O/config5_deep.txt
timeout 300 python tools/config5_probe.py 0.1 1e-3 6000 256 1024 backoff deepest > $O/config5_deep_backoff.txt 2>&1
echo "config5 deepest backoff rc=$?"; tail -22 $O/config5_deep_backoff.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt
