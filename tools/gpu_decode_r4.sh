#!/bin/bash
# Evidence run for the round-4 ART-V decode work (profiles/r04_decode_*): the hop probe, the persistent step against the five-launch form,
# its timeline, the step without its dependency chain, and bench.py --config 5 at batch 1 / 2 / 4 / 16.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
o=gpurun_out
timeout 300 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids > $o/decode_chain_probe.log
timeout 300 python tools/bench_decode_persistent.py 1 2 2>&1 | grep -v amdgpu.ids > $o/decode_persistent_vs_launches.log
MMVID_PD_NOWAIT=1 timeout 300 python tools/bench_decode_persistent.py 1 2>&1 | grep -v amdgpu.ids > $o/decode_persistent_nowait.log
timeout 300 python tools/decode_persistent_timeline.py 2>&1 | grep -v amdgpu.ids > $o/decode_persistent_timeline.log
for b in 1 2 4 16; do
  timeout 600 python bench.py --config 5 --batch $b --steps 2 --warmup 1 2>/dev/null | tail -1 > $o/bench_config5_b$b.json
done
MMVID_DECODE_PERSISTENT=0 timeout 600 python bench.py --config 5 --batch 1 --steps 2 --warmup 1 2>/dev/null | tail -1 > $o/bench_config5_b1_launches.json
# kernel statistics of the sampler at batch 1 (the persistent step) and batch 16 (the wide matrix-vector kernels)
export TMPDIR=/tmp
ROOT=$(pwd)
for b in 1 16; do
  rm -rf /tmp/prof_c5_$b
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5_$b -o c5 -- python $ROOT/bench.py --config 5 --batch $b --steps 1 --warmup 1 > /dev/null 2>&1)
  f=$(find /tmp/prof_c5_$b -name "*kernel_stats*" | head -1)
  [ -n "$f" ] && head -25 "$f" > $o/rocprofv3_config5_b${b}_kernel_stats.csv
done
