#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/trace
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace -o t -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --eager > /tmp/trace.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60] for r in rows]
# last step = last third; print every copyBuffer with its neighbours
idx = [i for i, n in enumerate(names) if 'copyBuffer' in n]
print('kernels', len(names), 'copyBuffer', len(idx))
seen = {}
for i in idx[len(idx) // 2:]:
    key = (names[i - 1], names[i + 1] if i + 1 < len(names) else '')
    seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, 'x  after', k[0], ' before', k[1])
mc = glob.glob('/tmp/trace/**/*memory_copy_trace.csv', recursive=True)
if mc:
    m = list(csv.DictReader(open(mc[0])))
    print('memory copies', len(m), m[0].keys() if m else '')
PY
