#!/bin/bash
# First device call of the next round.  Round 2 spent its GPU budget before these changes were made;
# they were verified on the CPU stand-in of the device table only:
#   * native bookkeeping of the prefix searches (include/ehm_search.h),
#   * barycentre witness, remembered optima, no phase one where feasibility is proven,
#   * suboptimality-test searches warm-started from the parent.
# Expected: tests green; the whole-box cell (52 737 nodes / 26 369 regions on the device when last
# run, 6.9 s, 1.82 M LPs) with the same tree, a few hundred thousand LPs, and a wall time the host
# bookkeeping dominates (about 2 s of it on the build container's CPU).
#   gpurun --timeout 1500 -- 'bash tools/next_round.sh'
mkdir -p gpurun_out
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_sequences.py -x -q -s > gpurun_out/next_sequences.txt 2>&1
echo "pytest test_gpu_sequences rc=$?" | tee -a gpurun_out/next_sequences.txt
timeout 300 python tools/_seq_probe.py 150000 frontier > gpurun_out/next_seq_probe.txt 2>&1
tail -3 gpurun_out/next_seq_probe.txt
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/next_bench.json 2> gpurun_out/next_bench.err
tail -c 600 gpurun_out/next_bench.json
# Then (a second call): tools/next_round/README.md -- apply midpoint_table.patch, rebuild, and run
#   python -m pytest tests/test_gpu_kernel_generations.py -k shared_midpoint -s
#   python -m pytest tests -m gpu -x -q ; python bench.py
