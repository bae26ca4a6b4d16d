export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for G in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQC_DCACHE_REQ_READ_1 SQC_DCACHE_REQ_READ_2 SQC_DCACHE_REQ_READ_4 SQC_DCACHE_REQ_READ_8 SQC_DCACHE_REQ_READ_16" "SQC_TC_DATA_READ_REQ SQC_DCACHE_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  timeout -k 10 200 rocprofv3 --pmc $G --output-format csv -d $R/gpurun_out/pmc_sqc/$(echo $G | cut -c1-20 | tr ' ' '_') -- python3 $R/bench.py --config ${CFG:-c3} --steps 1 --warmup 0 --no-cpu-baseline --no-host-visible > $R/gpurun_out/pmc_sqc.log 2>&1 || echo "FAIL $G"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_sqc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print("%-28s %.5g" % (k, tot[k]))
PY
