R=$PWD; O=$R/gpurun_out/prof_tr; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $B > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM --output-format csv -d $O/pmc4 -- $B > $O/pmc4.log 2>&1
python $R/tools/pmc_summary.py $O/summary.json $O/pmc3 $O/pmc4 > /dev/null
python - <<PY
import json
d=json.load(open("$O/summary.json"))["counters"]
for k in ("k2_lcss_decide","k2_lcss_expand"):
    c=d[k]; print(k, "FETCH MB/launch %.1f WRITE MB/launch %.1f VMEM insts %.3g scratch %s" % (c["FETCH_SIZE"]/c["_dispatches_pmc3"]/1024*2, c["WRITE_SIZE"]/c["_dispatches_pmc4"]/1024, c.get("SQ_INSTS_VMEM",0), c.get("_Scratch_Size")))
PY
find $O -name "*.csv" -delete
