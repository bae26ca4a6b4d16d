#!/bin/bash
# GPU box: with frames in flight the frame RATE of a share is bound by work, not by its dearest chain -- do split tiles (which
# cut the chain at 2.6x the tile's work) still pay?  One rank's share of C2 over {split parts} x {frames in flight}.
# usage: bash tools/exp_fif_split.sh <out.jsonl>
OUT=$1; : > "$OUT"
for n in 4 8; do for sp in 4 0; do for f in 1 2 3; do
  PTAMD_DEAL_SPLIT=$sp timeout -k 10 300 python3 bench.py --share-of $n --frames-in-flight $f --steps 16 --warmup 5 --no-cpu-baseline --config c2 2>/dev/null | tail -1 \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); d['split_parts']=$sp; print(json.dumps(d))" | tee -a "$OUT"
done; done; done
