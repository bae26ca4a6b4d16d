#!/bin/bash
# usage: bash tools/share_sweep.sh "<N values>" <bench args...>: kernel time of one rank's share of an N-GPU frame,
# raster order vs cost order
NS=$1; shift
for n in $NS; do for m in 0 1; do
  r=$(timeout -k 10 300 python bench.py --share-of $n --tile-schedule $m --steps 3 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % d['kernel_ms'])")
  echo "share-of $n tile-schedule $m $* -> $r"
done; done
