#!/bin/bash
# Round 5, first GPU call: the new kernels' tests, micro A/Bs, whole-step A/Bs, the bench line.  Everything under timeouts.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1; python oracle/build.py >> gpurun_out/build.log 2>&1
echo "== round-5 tests"; timeout 600 python -m pytest tests/test_round5_gpu.py -m gpu -x -q -s --timeout 200 -p no:cacheprovider > gpurun_out/t5.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t5.log; grep -E "stream|differ|FAILED|Error|error" gpurun_out/t5.log | head -30
echo "== attention / LN / GN micro"; timeout 300 python tools/r5_microbench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5_micro.log
echo "== attention legacy checks"; timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu | tee gpurun_out/attn.log
echo "== GEMM calls of a layer, stagger sweep"; timeout 300 python tools/bench_gemm_step.py 0 30 50 70 90 120 2>&1 | grep -v amdgpu | tee gpurun_out/gemm_step.log
echo "== conv layers"; timeout 300 python tools/conv_layer_profile.py 54 2>&1 | grep -v amdgpu > gpurun_out/conv_layers_54.log; head -30 gpurun_out/conv_layers_54.log; tail -3 gpurun_out/conv_layers_54.log
echo "== whole step A/B"; timeout 600 python tools/ab_multi.py base ln_fast=0 gn_fused=0 py:vae.stream=f32 attn_tail=0 attn_pk=0 gemm_stagger=50 gemm_stagger=80 "ln_fast=0,gn_fused=0,py:vae.stream=f32,attn_tail=0" 2>&1 | grep -v amdgpu | tee gpurun_out/ab_multi.log
echo "== attention + kernel tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_round4_gpu.py tests/test_round3_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attention or attn or layernorm or groupnorm or tower or vqgan or split" > gpurun_out/t_sel.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_sel.log; grep -E "^(FAILED|ERROR)" gpurun_out/t_sel.log | cut -c1-200
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-240 | tail -12
