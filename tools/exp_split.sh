#!/bin/bash
# GPU box: C4 / C3-SAH kernel time against the split of critically dear tiles.  usage: bash tools/exp_split.sh "<parts>" "<pcts>"
R=$GRAFT_REPO_ROOT; cd $R
run() { env PTAMD_SAH_SPLIT=$1 PTAMD_SPLIT_PCT=$2 timeout -k 10 300 python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-host-visible ${@:3} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % d['roofline']['kernel_ms_avg'])"; }
for s in $1; do for p in $2; do echo "split=$s pct=$p: c4 $(run $s $p --config c4)  c3-sah $(run $s $p --config c3 --bvh sah)"; done; done
