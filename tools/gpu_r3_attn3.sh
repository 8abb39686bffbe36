#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 600 python -m pytest tests/test_round3_gpu.py -x -q -k "resident or 12_layers" 2>&1 | tail -3 > gpurun_out/attn3_tests.log
for f in 0 1 2; do echo "attn_res=$f" >> gpurun_out/attn3_bench.log; MMVID_ATTN_RES=$f timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/attn3_bench.log; done
cat gpurun_out/attn3_tests.log gpurun_out/attn3_bench.log
