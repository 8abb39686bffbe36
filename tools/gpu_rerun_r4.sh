cd /root/repo; ROOT=$(pwd); export TMPDIR=/tmp
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-220
echo "== launcher, forced exchange"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/bench_ddp1.log 2> gpurun_out/bench_ddp1.err; grep "bench\]" gpurun_out/bench_ddp1.err | cut -c1-200; tail -3 gpurun_out/bench_ddp1.err
echo "== rocprofv3 kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --eager > $ROOT/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?")
find gpurun_out/prof -type f ! -name "*kernel_stats*" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --eager > $ROOT/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
  python tools/pmc_summary.py /tmp/pmc_$c gpurun_out/pmc_$c.csv
done
tail -3 gpurun_out/prof.log
