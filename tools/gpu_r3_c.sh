#!/bin/bash
# full GPU test suite + bench A/B through environment options
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python oracle/build.py >> gpurun_out/build.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
for v in $AB; do
  env $(echo $v | tr ',' ' ') timeout 300 python bench.py --steps 30 --no-cpu-baseline > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err; echo "$v: $(grep -o '\[bench\] [0-9.]* ms/step' gpurun_out/bench_ab.err)"
done
cp gpurun_out/bench_ab.json gpurun_out/bench_latest.json
