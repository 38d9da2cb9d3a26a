#!/bin/bash
# A/B of the inherited witness on the bench workload + full-size tree identity (gpurun)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ab_on.json 2> gpurun_out/ab_on.err
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --no-inherit-witness > gpurun_out/ab_off.json 2> gpurun_out/ab_off.err
python - <<'PY'
import json
for f in ('on', 'off'):
    try:
        j = json.loads(open('gpurun_out/ab_%s.json' % f).read().strip().splitlines()[-1])
        c = j['config']
        print(f, j['ms_per_step'], j['value'], 'lp', c.get('lp_solves_per_step'), 'inh', c.get('nodes_proved_open_by_inherited_witness_per_step'),
              'mid', c.get('nodes_proved_open_by_midpoint_per_step'), 'cert', c.get('leaves_closed_without_lp_per_step'),
              'it', c.get('mean_ipm_iterations'), 'margin', c.get('min_decision_margin'), 'frac', j['roofline']['frac'])
    except Exception as e:
        print(f, 'failed', e, open('gpurun_out/ab_%s.err' % f).read()[-2000:])
PY
PYTHONPATH=. python tools/fullsize_identity.py 2>&1 | tail -12
