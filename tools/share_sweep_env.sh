#!/bin/bash
# usage: bash tools/share_sweep_env.sh VAR "v1 v2" "<N values>" <bench args...>: kernel time of one rank's share per env value
VAR=$1; VALS=$2; NS=$3; shift 3
for v in $VALS; do for n in $NS; do
  r=$(env $VAR=$v timeout -k 10 300 python bench.py --share-of $n --steps 3 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['kernel_ms'])")
  echo "$VAR=$v share-of $n $* -> $r"
done; done
