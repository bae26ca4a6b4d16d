#!/bin/bash
# GPU box: correctness of the dealt walk, then C2 timing over {waves/SIMD build} x {deal_min}.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dealt or bit_exact or textures or depth_zero or random_scenes" > gpurun_out/deal_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/deal_tests.log
[ "$(tail -1 gpurun_out/deal_tests.log | grep -c passed)" = 1 ] || exit 1
L=multi-gpu-path-tracer_amd/libptamd.so; cp $L /tmp/lib_orig.so
: > gpurun_out/deal_sweep.txt
run() { timeout -k 10 120 python3 bench.py --config c2 --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))"; }
echo "mask walk (walk 2): $(run --walk 2)" | tee -a gpurun_out/deal_sweep.txt
for w in 8 7 6; do
  cp build/variants/libptamd_w$w.so $L
  for m in 0 24 32 40 48 56 65; do
    echo "waves $w deal_min $m: $(PTAMD_DEAL_MIN=$m run --walk 4)" | tee -a gpurun_out/deal_sweep.txt
  done
done
cp /tmp/lib_orig.so $L
