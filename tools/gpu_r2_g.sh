#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
for B in 1 4; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_dec$B -o dec -- python $ROOT/tools/bench_decode_step.py $B > $ROOT/gpurun_out/prof_dec$B.log 2>&1; echo "rocprof rc=$?")
f=$(find gpurun_out/prof_dec$B -name "*kernel_stats*" | head -1)
echo "== batch $B"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]
    print(f"{n:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f} max {float(r['MaxNs'])/1e3:7.2f}")
PY
find gpurun_out/prof_dec$B -type f ! -name "*kernel_stats*" -delete 2>/dev/null
done
