#!/bin/bash
# GPU box, repo root: one rank's share of an N-rank frame (N = 2, 4, 8) with 1 .. 4 frames in flight, C4 and C3 through the SAH
# tree, on the final round-4 kernels (bench.py --share-of; the single-GPU evidence behind DESIGN.md 6's projections).
# usage: bash tools/r04_shares.sh <out.jsonl>
OUT=$1; : > "$OUT"
run() {   # <label> <bench args...>
  L=$1; shift
  for n in 2 4 8; do
    for f in 1 2 3 4; do
      if [ $f = 4 ] && [ $n != 8 ]; then continue; fi
      timeout -k 10 300 python3 bench.py --share-of $n --frames-in-flight $f --steps 12 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | tail -1 \
        | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'config': '$L', 'n_ranks': d['n_ranks'], 'frames_in_flight': $f, 'ms_per_share': d['ms_per_share'], 'kernel_ms': d.get('kernel_ms'), 'segments': d['segments_per_share']}))" | tee -a "$OUT"
    done
  done
}
run c4 --config c4
run c3_sah --config c3 --bvh sah
run c2 --config c2
