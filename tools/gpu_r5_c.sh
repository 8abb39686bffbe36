#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1; python oracle/build.py >> gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
echo "== round-5 tests"; timeout 600 python -m pytest tests/test_round5_gpu.py -m gpu -q -s --timeout 200 -p no:cacheprovider > gpurun_out/t5.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t5.log; grep -E "stream|LN backward|FAILED|rror" gpurun_out/t5.log | head -40
echo "== replay probe"; timeout 300 python tools/bench_replay_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/replay_probe.log
echo "== micro"; timeout 300 python tools/r5_microbench.py attention 2>&1 | grep -v amdgpu | tee gpurun_out/r5_micro.log
echo "== whole step A/B"; timeout 600 python tools/ab_multi.py "attn_tail=4,attn_pk=0" "attn_tail=4,attn_pk=1" "attn_tail=4,attn_pk=2" "attn_tail=4,attn_pk=3" "attn_tail=6,attn_pk=3" "attn_tail=6,attn_pk=0" "attn_tail=0,attn_pk=0" 2>&1 | grep -v amdgpu | tee gpurun_out/ab_multi.log
echo "== all gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_gpu.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | cut -c1-220
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-240 | tail -5
