#!/bin/bash
# Run every GPU test in its own process (a device fault then costs one test, not the session) and log a summary.
# usage: tools/run_gpu_tests.sh [pytest -k expression]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOG=gpurun_out/tests.log
: > $LOG
rocm-smi --showproductname 2>/dev/null | head -8 >> $LOG
python -m mmvid_amd.build >> $LOG 2>&1
python oracle/build.py >> $LOG 2>&1
ids=$(python -m pytest tests -m gpu --collect-only -q ${1:+-k "$1"} 2>/dev/null | grep '::')
pass=0; fail=0
for t in $ids; do
  out=$(timeout 300 python -m pytest "$t" -x -q -s 2>&1)
  rc=$?
  if [ $rc -eq 0 ]; then pass=$((pass+1)); echo "PASS $t" >> $LOG; echo "$out" | grep -E "relerr|match|loss|norm" | sed 's/^/     /' >> $LOG
  else fail=$((fail+1)); echo "FAIL($rc) $t" >> $LOG; echo "$out" | tail -40 | sed 's/^/     /' >> $LOG; fi
done
echo "SUMMARY pass=$pass fail=$fail" | tee -a $LOG
grep -E "^(PASS|FAIL)" $LOG | sort | uniq -c | sort -rn | head -3 > /dev/null
grep -E "^FAIL" $LOG
