#!/bin/bash
# resident attention kernels: parity with the streaming form, microbench, whole-step A/B
mkdir -p gpurun_out; cd /root/repo
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py -x -q -k "attention or attn" 2>&1 | tail -8 > gpurun_out/attnres_tests.log
for f in 0 1; do echo "attn_res=$f" >> gpurun_out/attnres_bench.log; MMVID_ATTN_RES=$f timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/attnres_bench.log; done
timeout 600 python tools/ab_graph.py attn_res 0 1 2>&1 | tail -12 > gpurun_out/attnres_ab.log
cat gpurun_out/attnres_tests.log gpurun_out/attnres_bench.log gpurun_out/attnres_ab.log
