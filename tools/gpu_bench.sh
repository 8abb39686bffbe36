#!/bin/bash
# smoke + bench + rocprofv3 kernel stats on the GPU box; everything lands in gpurun_out/
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== smoke" ; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py ${BENCH_ARGS:---steps 5 --warmup 2} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
if [ -z "$NO_PROF" ]; then
echo "== rocprofv3 kernel stats"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?")
find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the small summaries
find gpurun_out/prof -type f ! -name "*stats*" -delete
fi
