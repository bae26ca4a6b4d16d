#!/bin/bash
# GPU box: A/B of the SAH kernel over an environment switch.  usage: VAR=<env name> VALS="a b c" bash tools/exp_c4.sh
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || python3 -m pytest tests/test_gpu_sah.py tests/test_gpu_c4.py -m gpu -x -q 2>&1 | tail -2
: > gpurun_out/exp_c4.txt
for v in ${VALS:-0}; do
  for cfg in "--config c4" "--config c3 --bvh sah"; do
    r=$(env $VAR=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-visible $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms per frame (%d in flight)  %s' % (d['ms_per_step'], d['frames_in_flight'], d['roofline']['kernel']))")
    echo "$cfg $VAR=$v: $r" | tee -a gpurun_out/exp_c4.txt
  done
done
