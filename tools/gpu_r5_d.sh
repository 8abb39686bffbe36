#!/bin/bash
# Round-5 checkpoint run: the pruned tree (one code path per kernel) -- all GPU tests, then the same-box comparison of the training step
# against the round-4 tree (a git worktree of 63f6de5 built under _r04/, alternating runs so that box drift shows), kernel timings.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1; python oracle/build.py >> gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
echo "== all gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_gpu.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | cut -c1-220
echo "== same-box r04 vs r05"
for rep in 1 2; do
  (cd _r04 && timeout 300 python bench.py --no-cpu-baseline --no-exact --steps 50 2>&1 | grep "bench\]" | tail -1 | sed "s/^/r04 tree (63f6de5) run $rep: /") | tee -a gpurun_out/same_box.log
  timeout 300 python bench.py --no-cpu-baseline --no-exact --steps 50 2>&1 | grep "bench\]" | tail -1 | sed "s/^/r05 tree run $rep: /" | tee -a gpurun_out/same_box.log
done
echo "== micro"; timeout 300 python tools/microbench.py 2>&1 | grep -v amdgpu | tee gpurun_out/micro.log
echo "== gemm layer calls"; timeout 300 python tools/bench_gemm_step.py 2>&1 | grep -v amdgpu | tee gpurun_out/gemm_step.log
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-240 | tail -5
