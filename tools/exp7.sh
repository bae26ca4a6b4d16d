#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r02c_c4/WRITE_SIZE -- python3 $R/bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-host-visible > $R/gpurun_out/pmc_r02c_c4_WRITE_SIZE.log 2>&1 || echo FAIL
cd $R && python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_r02c_c4/WRITE_SIZE/**/*_counter_collection.csv", recursive=True):
    v = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"])
    print("C4 WRITE_SIZE MB", v * 1024 / 1e6)
PY
