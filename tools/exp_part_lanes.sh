#!/bin/bash
# GPU box: A/B of parts of split tiles walking by lanes (part_walk_by_lanes, PTAMD_PART_LANES) on one rank's share of a
# C2 / C5 frame, against the same build with it off and against a baseline build (build/variants/<base>.so, optional).
# usage: bash tools/exp_part_lanes.sh <out.jsonl> [base.so]
OUT=$1; BASE=$2; : > "$OUT"
one() {   # <label> <n> <bench args...>
  L=$1; N=$2; shift 2
  timeout -k 10 300 python3 bench.py --share-of $N --steps 5 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'variant': '$L', 'n_ranks': d['n_ranks'], 'kernel_ms': d['kernel_ms']}))" | tee -a "$OUT"
}
for cfg in c2 c5; do
  for n in 2 4 8; do
    PTAMD_PART_LANES=1 one "$cfg by-lanes" $n --config $cfg
    PTAMD_PART_LANES=0 one "$cfg off" $n --config $cfg
    [ -n "$BASE" ] && PTAMD_LIBRARY=$(readlink -f $BASE) one "$cfg base" $n --config $cfg
  done
done
