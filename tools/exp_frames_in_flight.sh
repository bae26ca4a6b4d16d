#!/bin/bash
# GPU box: one rank's share of a frame with 1 / 2 / 3 frames in flight (each in its own framebuffer on its own stream; a
# schedule per framebuffer): ms per share in the steady state, i.e. how much of the wave slots that a share's last chains
# leave idle the next frame's waves can use.  usage: bash tools/exp_frames_in_flight.sh <out.jsonl> <label> [bench args, e.g. --config c3 --bvh sah]
OUT=$1; L=$2; shift 2; : > "$OUT"
for n in 2 4 8; do
  for f in 1 2 3; do
    timeout -k 10 300 python3 bench.py --share-of $n --frames-in-flight $f --steps 12 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | tail -1 \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); d['config']='$L'; print(json.dumps(d))" | tee -a "$OUT"
  done
done
