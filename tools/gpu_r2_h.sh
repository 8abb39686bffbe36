#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "strip or vqgan or conv or vid_negative or groupnorm or bert_training" > gpurun_out/sel.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/sel.log | cut -c1-250
grep -E "^E  " gpurun_out/sel.log | head -12 | cut -c1-250
echo "== conv profile"; timeout 300 python tools/conv_layer_profile.py 54 > gpurun_out/conv_profile_54.log 2>&1; grep -E "conv   m0 (128x128|64x64|32x32)|total" gpurun_out/conv_profile_54.log | cut -c1-160
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
for k in d['kernels']: print('    ',k['kernel'],round(k['ms_per_step'],3),'ms',round(k['tflops'],1),'TF')
PY
echo "== decode floor"; timeout 300 python tools/bench_decode_step.py 1 2>&1 | tail -4 | cut -c1-200
