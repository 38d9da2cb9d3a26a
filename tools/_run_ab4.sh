#!/bin/bash
# A/B of the inherited witness on the wide-kernel workload (config 4) + its parity tests (gpurun)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --no-cpu-baseline --workload config4 --steps 2 --warmup 1 > gpurun_out/ab4_on.json 2> gpurun_out/ab4_on.err
python bench.py --no-cpu-baseline --workload config4 --steps 2 --warmup 1 --no-inherit-witness > gpurun_out/ab4_off.json 2> gpurun_out/ab4_off.err
python - <<'PY'
import json
for f in ('on', 'off'):
    try:
        j = json.loads(open('gpurun_out/ab4_%s.json' % f).read().strip().splitlines()[-1])
        c = j['config']
        print(f, j['ms_per_step'], j['value'], 'lp', c.get('lp_solves_per_step'), 'inh', c.get('nodes_proved_open_by_inherited_witness_per_step'),
              'cert', c.get('leaves_closed_without_lp_per_step'), 'nodes', c.get('nodes_per_step'), 'regions', c.get('regions_per_step'),
              'it', c.get('mean_ipm_iterations'), 'margin', c.get('min_decision_margin'), 'frac', j['roofline']['frac'])
    except Exception as e:
        print(f, 'failed', e, open('gpurun_out/ab4_%s.err' % f).read()[-2000:])
PY
python -m pytest tests/test_gpu_wide.py -q -m gpu 2>&1 | tail -4
