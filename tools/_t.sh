cd /root/repo
export TMPDIR=/tmp
rm -rf /tmp/p64
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p64 -o dec -- python /root/repo/bench.py --config 5 --batch 64 --steps 1 --warmup 1 > /tmp/p64.log 2>&1; echo "rocprof rc=$?")
f=$(find /tmp/p64 -name "*kernel_stats*" | head -1); cp $f gpurun_out/dec64_kernel_stats.csv
