#!/bin/bash
# Full evidence run on the GPU box -> gpurun_out/ : bench (with the CPU baseline) FIRST, on the fresh box, then
# smoke, all GPU tests (one pytest process, verbose), rocprofv3 kernel stats of the bench command, and two PMC
# passes (FETCH_SIZE, WRITE_SIZE; counters only, kernel-trace) reduced to per-kernel averages.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
python oracle/build.py >> gpurun_out/build.log 2>&1
rocm-smi --showproductname --showmeminfo vram 2>/dev/null | head -20 > gpurun_out/rocm_smi.txt
nproc > gpurun_out/host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/host.txt
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "bench\]" gpurun_out/bench.err
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -v -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
echo "== rocprofv3 kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?")
find gpurun_out/prof -type f ! -name "*stats*" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
  python tools/pmc_summary.py /tmp/pmc_$c gpurun_out/pmc_$c.csv
done
