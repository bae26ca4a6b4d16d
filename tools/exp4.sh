#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/exp4_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/exp4_tests.log
for algo in fsfl dsfl; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --rehearse --steps 3 --warmup 3 --config c2 --algo $algo --no-host-visible > gpurun_out/exp4_rehearse_$algo.log 2>&1; echo "rehearse $algo rc=$?"; tail -1 gpurun_out/exp4_rehearse_$algo.log | cut -c1-700
done
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --rehearse --steps 2 --warmup 3 --config c5 --spp 64 > gpurun_out/exp4_rehearse_c5.log 2>&1; echo "rehearse c5 rc=$?"; tail -1 gpurun_out/exp4_rehearse_c5.log | cut -c1-900
