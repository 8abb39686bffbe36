#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python oracle/build.py >> gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -q -m gpu -x -k "$1" > gpurun_out/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_k.log
