#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
run() { timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))"; }
: > gpurun_out/exp_c4.txt
for pf in 0 1; do
  echo "c4 prefetch $pf: $(PTAMD_SAH_PREFETCH=$pf run --config c4)" | tee -a gpurun_out/exp_c4.txt
  echo "c3 sah prefetch $pf: $(PTAMD_SAH_PREFETCH=$pf run --config c3 --bvh sah)" | tee -a gpurun_out/exp_c4.txt
done
PTAMD_SAH_PREFETCH=1 python3 -m pytest tests/test_gpu_sah.py tests/test_gpu_c4.py -m gpu -x -q 2>&1 | tail -2
