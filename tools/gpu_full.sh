#!/bin/bash
# Full evidence run on the GPU box: all GPU tests, smoke, bench (with CPU baseline), rocprofv3 kernel stats,
# and two PMC passes (FETCH_SIZE, WRITE_SIZE) reduced to per-kernel averages.  Output: gpurun_out/
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/run_gpu_tests.sh
BENCH_ARGS="--steps 10 --warmup 3" bash tools/gpu_bench.sh
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
  python tools/pmc_summary.py /tmp/pmc_$c gpurun_out/pmc_$c.csv
done
rocm-smi --showproductname --showmeminfo vram 2>/dev/null | head -20 > gpurun_out/rocm_smi.txt
nproc > gpurun_out/host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/host.txt
