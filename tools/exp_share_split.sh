#!/bin/bash
# GPU box: one rank's share of a C2 frame over {waves per split tile} x {split ratio}: how the share's time depends on what is split.
# usage: bash tools/exp_share_split.sh [N=8] ["parts ratio" ...]
N=${1:-8}; shift
[ $# -eq 0 ] && set -- "4 65" "4 55" "4 75" "8 75" "8 85" "8 92" "2 65"
for cfg in "$@"; do set -- $cfg; r=$(PTAMD_DEAL_SPLIT=$1 PTAMD_SPLIT_RATIO=$2 timeout -k 10 300 python bench.py --share-of $N --steps 4 --warmup 3 --no-cpu-baseline --config c2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['kernel_ms'])"); echo "1/$N parts $1 ratio $2 -> $r"; done
