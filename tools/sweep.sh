#!/bin/bash
# usage: bash tools/sweep.sh VAR "v1 v2 ..." <bench args...> ; prints ms_per_step per value
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  ms=$(env $VAR=$v timeout -k 10 300 python bench.py "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.1f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$VAR=$v $* -> $ms ms"
done
