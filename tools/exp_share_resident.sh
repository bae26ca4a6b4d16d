#!/bin/bash
# GPU box: one rank's share of a C2 frame over resident workgroups per CU (PTAMD_RESIDENT) x split ratio: does a share whose
# time is its dearest parts' chain gain from fewer neighbours per SIMD?
# usage: bash tools/exp_share_resident.sh [N=8] ["resident ratio" ...]
N=${1:-8}; shift
[ $# -eq 0 ] && set -- "7 65" "6 65" "5 65" "4 65" "3 65"
for cfg in "$@"; do set -- $cfg; r=$(PTAMD_RESIDENT=$1 PTAMD_SPLIT_RATIO=$2 timeout -k 10 300 python bench.py --share-of $N --steps 4 --warmup 3 --no-cpu-baseline --config c2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['kernel_ms'])"); echo "1/$N resident $1 ratio $2 -> $r"; done
