#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
for g in 0 1; do
  rm -rf /tmp/pmc_g$g
  (cd /tmp && MMVID_GEMM_GROUPN=$g timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_g$g -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --eager > /tmp/pmc_g$g.log 2>&1)
  python tools/pmc_summary.py /tmp/pmc_g$g gpurun_out/pmc_fetch_groupn$g.csv > /dev/null
  echo "groupn=$g"; grep -E "gemm_bf16_lw_kernel<false, false, [23]>|gemm_bf16_lw_kernel<false, true, 4>" gpurun_out/pmc_fetch_groupn$g.csv | cut -d, -f1,3,4
done
for g in 0 1; do echo "groupn=$g:"; MMVID_GEMM_GROUPN=$g timeout 300 python tools/bench_gemm_epi.py 2>&1 | grep -E "qkv fwd|fc fwd|dX proj" | cut -c1-200; done
