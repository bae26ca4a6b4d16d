#!/bin/bash
# GPU box: C2 timing of the dealt walk over {waves/SIMD build} x {deal_min} (variants prebuilt under build/variants/).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
# variants are loaded through PTAMD_LIBRARY: the product library stays as it is
: > gpurun_out/deal_sweep.txt
run() { timeout -k 10 120 python3 bench.py --config c2 --steps 4 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))"; }
echo "mask walk (walk 2): $(run --walk 2)" | tee -a gpurun_out/deal_sweep.txt
for w in ${WAVES:-8 7}; do
  export PTAMD_LIBRARY=$R/build/variants/libptamd_w$w.so
  for m in ${MINS:-32 40 48 56 60}; do
    echo "waves $w deal_min $m: $(PTAMD_DEAL_MIN=$m run --walk 4)" | tee -a gpurun_out/deal_sweep.txt
  done
done
unset PTAMD_LIBRARY
