#!/bin/bash
# On the GPU box, from the repo root: bash tools/shares_json.sh <out.jsonl>
# Kernel time of ONE rank's share of an N-rank frame (N = 1, 2, 4, 8; bench.py --share-of, cost-ordered tiles)
# for C2, C3 (SAH tree) and C5: the single-GPU evidence behind the strong-scaling projections in DESIGN.md 6.
OUT=$1; : > "$OUT"
run() {   # <label> <bench args...>
  L=$1; shift
  for n in 1 2 4 8; do
    if [ $n = 1 ]; then
      line=$(timeout -k 10 600 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 \
             | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'config': '$L', 'n_ranks': 1, 'kernel_ms': d['roofline']['kernel_ms_avg'], 'ms': d['ms_per_step'], 'segments': d['segments_per_frame']}))")
    else
      line=$(timeout -k 10 600 python3 bench.py --share-of $n --steps 3 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 \
             | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'config': '$L', 'n_ranks': d['n_ranks'], 'kernel_ms': d['kernel_ms'], 'ms': d['ms_per_share'], 'segments': d['segments_per_share']}))")
    fi
    echo "$line" | tee -a "$OUT"
  done
}
run c2 --config c2
run c3_sah --config c3 --bvh sah
run c5 --config c5
run c3_ref --config c3
