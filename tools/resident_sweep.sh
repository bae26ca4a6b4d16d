#!/bin/bash
# GPU box: kernel time against resident workgroups per CU (PTAMD_RESIDENT) for the persistent kernels -- how much of the
# frame time is latency that more waves would hide.  usage: bash tools/resident_sweep.sh <tag> "<bench args>" "<caps>"
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; TAG=${1:-c4}; ARGS=${2:---config c4}; CAPS=${3:-1 2 3 4 5 6 7 8}
OUT=gpurun_out/resident_$TAG.txt; : > $OUT
for c in $CAPS; do
  line=$(PTAMD_RESIDENT=$c timeout -k 10 300 python3 bench.py $ARGS --steps 4 --warmup 2 --no-cpu-baseline --no-host-visible 2>/dev/null | tail -1)
  echo "resident $c: $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))")" | tee -a $OUT
done
