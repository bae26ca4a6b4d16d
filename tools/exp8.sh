#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/exp8_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/exp8_tests.log
python3 tools/fuzz_parity.py 400000 6000 > gpurun_out/fuzz_r02d.log 2>&1; tail -1 gpurun_out/fuzz_r02d.log
run() { timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))"; }
echo "c2: $(run --config c2)"; echo "c2: $(run --config c2)"; echo "c5: $(run --config c5 --steps 2)"
