#!/bin/bash
# GPU box: one rank's eighth of a C2 frame over {waves per split tile} x {split ratio}: how the share's time depends on what is split.
for cfg in "4 65" "4 55" "4 75" "8 75" "8 85" "8 92" "2 65"; do set -- $cfg; r=$(PTAMD_DEAL_SPLIT=$1 PTAMD_SPLIT_RATIO=$2 timeout -k 10 300 python bench.py --share-of 8 --steps 4 --warmup 3 --no-cpu-baseline --config c2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['kernel_ms'])"); echo "parts $1 ratio $2 -> $r"; done
