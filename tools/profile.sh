#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box: gpurun -- 'bash tools/profile.sh TAG').
# Kernel-trace statistics of the full bench workload, then separate --pmc passes (SQ issue /
# wait counters, LDS, HBM bytes) over one step of the same workload.
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# BENCH_ARGS: e.g. "--workload config4" (the wide kernels); the MFMA pass only matters there
B="python $R/bench.py --no-cpu-baseline --no-secondary ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 3 --warmup 1 > $O/stats.log 2>&1
S="--steps 1 --warmup 0 ${PROFILE_ARGS:-}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc1 -- $B $S > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $O/pmc2 -- $B $S > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $B $S > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -- $B $S > $O/pmc4.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc5 -- $B $S > $O/pmc5.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/prof_${TAG}_summary.json $O/stats $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 > /dev/null
# keep the merged-back payload small: the raw per-dispatch CSVs stay on the box
find $O -name "*counter_collection.csv" -size +2M -delete
tail -2 $O/stats.log | cut -c1-300
