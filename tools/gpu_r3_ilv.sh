#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 600 python -m pytest tests/test_round3_gpu.py -x -q -k "interleaved or resident or split_k" 2>&1 | tail -2 > gpurun_out/ilv_tests.log
for f in 0 8; do echo "attn_occ=$f" >> gpurun_out/ilv_bench.log; MMVID_ATTN_OCC=$f timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/ilv_bench.log; done
timeout 600 python tools/ab_graph.py attn_occ 0 8 2>&1 | tail -2 >> gpurun_out/ilv_bench.log
cat gpurun_out/ilv_tests.log gpurun_out/ilv_bench.log
