#!/bin/bash
# A/B of the graph-replayed step (default) against the eagerly launched step (--eager) under host contention: N busy Python loops compete with the benchmark's host thread.
cd "$(dirname "$0")/.."
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
N=${1:-32}
pids=""
for i in $(seq $N); do python -c "
import time
t=time.time()
while time.time()-t < 200: sum(range(10000))" & pids="$pids $!"; done
sleep 2
for v in "" "--eager" "" "--eager"; do
  echo "== bench.py $v with $N busy processes"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $v 2>&1 | grep "bench\]"
done
kill $pids 2>/dev/null
wait 2>/dev/null
