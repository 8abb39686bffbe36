#!/bin/bash
# Round-2 first GPU pass: new parity tests (verbose), old suite, smoke, short bench.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
python oracle/build.py >> gpurun_out/build.log 2>&1
echo "== parity tests"; timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -v -s --timeout 400 -p no:cacheprovider > gpurun_out/parity.log 2>&1; echo "parity rc=$?"
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/parity.log | tail -45
echo "== old suite"; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/old.log 2>&1; echo "old rc=$?"; tail -15 gpurun_out/old.log
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err; cut -c1-600 gpurun_out/bench.log
