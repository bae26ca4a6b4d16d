#!/bin/bash
# GPU box: sweep one environment switch of the SAH kernel on C4 and C3-SAH.  usage: VAR=<name> VALS="a b c" bash tools/exp_pct.sh
R=$GRAFT_REPO_ROOT; cd $R
run() { env $VAR=$1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible ${@:2} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % d['roofline']['kernel_ms_avg'])"; }
for v in $VALS; do echo "$VAR=$v: c4 $(run $v --config c4)  c3-sah $(run $v --config c3 --bvh sah)"; done
