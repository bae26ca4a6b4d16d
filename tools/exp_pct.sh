#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { env PTAMD_SPEC_PCT=$1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible ${@:2} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % d['roofline']['kernel_ms_avg'])"; }
for v in 75 85 90 95 99; do echo "c4 pct $v: $(run $v --config c4)"; done
for v in 30 40 45; do echo "c3 pct $v: $(run $v --config c3 --bvh sah)"; done
