#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "split_k or gemm or dw or tower or graphed or reproducible" 2>&1 | tail -3 > gpurun_out/fr_tests.log
timeout 600 python tools/ab_graph.py gemm_fused_reduce 0 1 2>&1 | tail -4 > gpurun_out/fr_ab.log
cat gpurun_out/fr_tests.log gpurun_out/fr_ab.log
