#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=multi-gpu-path-tracer_amd/libptamd.so; cp $L /tmp/lib_orig.so
run() { timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms' % (d['roofline']['kernel_ms_avg']))"; }
: > gpurun_out/exp3.txt
for b in 4 2 1; do
  cp build/variants/libptamd_b$b.so $L
  for lr in 0 1; do
    echo "waves/block $b lrpt $lr: 1080 $(PTAMD_LRPT=$lr run --config c2)  2160 $(PTAMD_LRPT=$lr run --config c2 --height 2160)  share8 $(PTAMD_LRPT=$lr timeout -k 10 300 python3 bench.py --config c2 --share-of 8 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.2f' % json.loads(sys.stdin.readline())['kernel_ms'])")" | tee -a gpurun_out/exp3.txt
  done
done
cp /tmp/lib_orig.so $L
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dealt or bit_exact" 2>&1 | tail -2
