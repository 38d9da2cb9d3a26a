cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r1 | head -20
cat $GRAFT_REPO_ROOT/gpurun_out/prof_r1/*kernel_stats.csv | head -20
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r1a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --abs-frac 0.05 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_a.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r1b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --abs-frac 0.05 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_b.err
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_r1a $GRAFT_REPO_ROOT/gpurun_out/pmc_r1b
