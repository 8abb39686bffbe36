#!/bin/bash
# Round-6 session A: flip census of the tokeniser modes, per-op encoder profile in the bf16 and exact-index modes, baseline bench of this box.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
python -m mmvid_amd.build > $o/build.log 2>&1; python oracle/build.py >> $o/build.log 2>&1
echo "== census"; timeout 900 python tools/flip_census.py 2>&1 | grep -v -i warn > $o/flip_census.log; tail -12 $o/flip_census.log | cut -c1-260
for m in bf16 mixed split; do timeout 300 python tools/conv_layer_profile.py 54 $m 2>&1 | grep -v amdgpu > $o/conv_layers_54_$m.log; tail -5 $o/conv_layers_54_$m.log; done
echo "== bench"; timeout 900 python bench.py > $o/bench.log 2> $o/bench.err; echo "bench rc=$?"; grep "bench\]" $o/bench.err | cut -c1-220
