#!/bin/bash
# A/B of the witnesses on the bench and config-4 workloads, full-size identity, the tests that pin them
cd ${GRAFT_REPO_ROOT:-.}
bash tools/_run_ab.sh
bash tools/_run_ab4.sh
python -m pytest tests/test_gpu_kernel_generations.py tests/test_gpu_bench_parity.py tests/test_gpu_partition.py tests/test_gpu_rebalance.py -q -m gpu 2>&1 | tail -3
