#!/bin/bash
# full GPU test suite + bench (A/B of gemm_epi through the environment)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python oracle/build.py >> gpurun_out/build.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
for e in 0 1 0 1; do
  MMVID_GEMM_EPI=$e timeout 300 python bench.py --steps 30 --no-cpu-baseline > gpurun_out/bench_epi$e.json 2> gpurun_out/bench_epi$e.err; echo "epi $e: $(grep -o '\[bench\] [0-9.]* ms/step' gpurun_out/bench_epi$e.err)"
done
