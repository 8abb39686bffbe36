#!/bin/bash
# graph-replay stress of the config-2 / config-4 step on the round-3 kernels (loader-wave GEMM, new epilogues): every loss,
# gradient and parameter must stay finite (the round-2 LDS-DMA race showed as NaN within tens of replays)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
(for s in 42 44 46; do timeout 500 python tools/stress_nan2.py 300 $s graph 2>&1 | grep -E "finite|step |loss" | tail -3; done
 for s in 51 52; do timeout 500 python tools/stress_nan2.py 200 $s graph 4 2>&1 | grep -E "finite|step |loss" | tail -3; done) > gpurun_out/stress_nan.log 2>&1
cat gpurun_out/stress_nan.log | cut -c1-220
