#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
python oracle/build.py >> gpurun_out/build.log 2>&1
echo "== selected tests"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_models_gpu.py -m gpu -q --timeout 400 -p no:cacheprovider \
  -k "frontend or artv or vid_negative or config or graphed or trains" > gpurun_out/sel.log 2>&1; echo "sel rc=$?"
tail -5 gpurun_out/sel.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/sel.log | cut -c1-200
echo "== conv profile"; timeout 300 python tools/conv_layer_profile.py 54 > gpurun_out/conv_profile_54.log 2>&1; tail -45 gpurun_out/conv_profile_54.log | cut -c1-200
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/bench.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
    print('ms/step',d['ms_per_step'],'value',d['value'],'launch',d['config']['step_launch'])
    for k in d['kernels']: print('  ',k['kernel'],round(k['ms_per_step'],3),'ms',round(k['tflops'],1),'TF',k['launches_per_step'])
except Exception as e: print('bench parse',e)
PY
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof -o bench -- python $OLDPWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager > $OLDPWD/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?")
find gpurun_out/prof -type f ! -name "*kernel_stats*" -delete 2>/dev/null
f=$(find gpurun_out/prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-160
