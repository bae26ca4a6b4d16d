#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/exp6_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/exp6_tests.log
python3 tools/fuzz_parity.py 300000 8000 > gpurun_out/fuzz_r02c.log 2>&1; tail -1 gpurun_out/fuzz_r02c.log
run() { timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))"; }
echo "c2: $(run --config c2)"; echo "c3 sah: $(run --config c3 --bvh sah)"; echo "c4: $(run --config c4)"
export TMPDIR=/tmp; cd /tmp
for C in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_r02c_c2/$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-visible > $R/gpurun_out/pmc_r02c_c2_$C.log 2>&1 || echo FAIL $C
done
cd $R && python3 - <<'PY'
import csv, glob
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob("gpurun_out/pmc_r02c_c2/%s/**/*_counter_collection.csv" % c, recursive=True):
        v = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"])
        print(c, "KiB:", v, "=> MB", v * 1024 / 1e6)
PY
