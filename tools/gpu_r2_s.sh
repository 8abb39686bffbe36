#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
for sp in 1 0 1 0; do echo "== launcher, forced exchange, sparse=$sp"; MMVID_SPARSE_TABLES=$sp timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/bench_ddp1.log 2> gpurun_out/bench_ddp1.err; grep "bench\]" gpurun_out/bench_ddp1.err | cut -c1-200; done
