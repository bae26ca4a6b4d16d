#!/bin/bash
# usage: bash tools/sweep_arg.sh "--flag" "v1 v2 ..." <bench args...> ; prints ms_per_step per value of the flag
FLAG=$1; VALS=$2; shift 2
for v in $VALS; do
  ms=$(timeout -k 10 300 python bench.py $FLAG $v "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f ms (kernel %.1f)' % (d['ms_per_step'], d['roofline']['kernel_ms_avg']))")
  echo "$FLAG $v $* -> $ms"
done
