#!/bin/bash
# GPU box, repo root: vector-memory-path counters of one bench launch, one rocprofv3 --pmc pass per group
# (groups that the hardware cannot schedule together simply fail and are reported; every pass has its own timeout: a pass
# with TD_* counters hung a whole run on this pool and those are no longer asked for).  usage: bash tools/pmc_mem.sh <tag> [bench args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; TAG=$1; shift
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-host-visible $@"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  echo "group $i: $GROUP"
  timeout -k 10 150 rocprofv3 --pmc $GROUP --output-format csv -d $R/gpurun_out/pmcmem_$TAG/g$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmcmem_${TAG}_g$i.log 2>&1 || echo "FAIL group $i: $GROUP"
done <<'GROUPS'
TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum
GROUPS
cd $R && python3 - $TAG <<'PY'
import csv, glob, sys, collections, json
tag = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmcmem_%s/*/**/*_counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(dict(tot), open("gpurun_out/pmcmem_%s.json" % tag, "w"), indent=1)
for k in sorted(tot): print("%-44s %.5g" % (k, tot[k]))
PY
