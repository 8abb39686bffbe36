#!/bin/bash
# round 3, call A: the new parity tests + the front-end tests (state buffer changed) + baseline bench + strict bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
python oracle/build.py >> gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests/test_round3_gpu.py -q -s -m gpu > gpurun_out/r3_tests.log 2>&1; echo "r3 tests rc=$?" 
tail -5 gpurun_out/r3_tests.log
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "frontend or vid_negative or device_frontend or state_dict or graphed" > gpurun_out/r3_frontend_tests.log 2>&1; echo "frontend tests rc=$?"
tail -3 gpurun_out/r3_frontend_tests.log
timeout 300 python bench.py --steps 30 --no-cpu-baseline > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err; echo "bench rc=$?"
tail -2 gpurun_out/bench_base.err
timeout 300 python bench.py --steps 10 --strict --no-cpu-baseline > gpurun_out/bench_strict.json 2> gpurun_out/bench_strict.err; echo "strict rc=$?"
tail -2 gpurun_out/bench_strict.err
