#!/bin/bash
# First GPU call of the next round: validate and measure what round 1 left unvalidated.
#   gpurun --timeout 900 -- 'bash tools/next_round.sh'
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
# 1. the midpoint-first flow of the persistent kernel (option "mid_first"): identical tree?
EHM_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_kernel_generations.py -q -x \
    -k midpoint_first > gpurun_out/mid_first_test.log 2>&1; tail -5 gpurun_out/mid_first_test.log
# 2. its effect on the bench workload (expected: ~154 -> ~115 ms per partition)
for F in "" "--mid-first"; do
  python bench.py --no-cpu-baseline $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('bench $F', d['value'], d['regions_per_s'], d['ms_per_step'], c['lp_solves_per_step'], c['leaves_closed_without_lp_per_step'], c['nodes_proved_open_by_midpoint_per_step'], c['nodes_per_step'])"
done
# 3. full-size identity of the tree with the option on (against the sweeps at full accuracy)
EHM_MID_FIRST=1 PYTHONPATH=. timeout 200 python tools/_sign_stop_check.py 2>&1 | tail -2
