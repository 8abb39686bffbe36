#!/bin/bash
# Round-4 evidence run on the GPU box -> gpurun_out/ : headline bench (with the CPU baseline) FIRST on the fresh box, smoke, all
# GPU tests (one pytest process), strict / config 4 / config 5 / sampling bench lines, the launcher path with the gradient exchange
# forced, rocprofv3 kernel stats of the bench command, two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only), GEMM timeline.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
python oracle/build.py >> gpurun_out/build.log 2>&1
rocm-smi --showproductname --showmeminfo vram 2>/dev/null | head -20 > gpurun_out/rocm_smi.txt
nproc > gpurun_out/host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/host.txt
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-220
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log | cut -c1-220
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -v -s --timeout 400 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/pytest_gpu.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | cut -c1-200
echo "== strict"; timeout 600 python bench.py --strict --steps 20 --no-cpu-baseline > gpurun_out/bench_strict.log 2> gpurun_out/bench_strict.err; grep "bench\]" gpurun_out/bench_strict.err | cut -c1-200
echo "== strict split"; timeout 600 python bench.py --strict split --steps 20 --no-cpu-baseline --no-exact > gpurun_out/bench_split.log 2> gpurun_out/bench_split.err; grep "bench\]" gpurun_out/bench_split.err | cut -c1-200
echo "== config 4"; timeout 600 python bench.py --config 4 --steps 20 --warmup 3 > gpurun_out/bench_c4.log 2> gpurun_out/bench_c4.err; grep "bench\]" gpurun_out/bench_c4.err | cut -c1-200
echo "== config 4, the recipe batch on one GPU"; timeout 600 python bench.py --config 4 --batch 20 --steps 10 --warmup 3 --no-cpu-baseline --no-exact > gpurun_out/bench_c4_b20.log 2> gpurun_out/bench_c4_b20.err; grep "bench\]" gpurun_out/bench_c4_b20.err | cut -c1-200
echo "== config 5"; timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/bench_c5.log 2> gpurun_out/bench_c5.err; echo "rc=$?"
echo "== BERT sampling (mask-predict)"; timeout 900 python bench.py --sample --steps 3 --warmup 1 > gpurun_out/bench_bert_sampling.log 2> gpurun_out/bench_bert_sampling.err; grep "bench\]" gpurun_out/bench_bert_sampling.err | cut -c1-200
echo "== launcher, forced exchange"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/bench_ddp1.log 2> gpurun_out/bench_ddp1.err; grep "bench\]" gpurun_out/bench_ddp1.err | cut -c1-200
echo "== self-launch on a 1-GPU box"; timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/bench_gpus2.log 2> gpurun_out/bench_gpus2.err; echo "rc=$? (expected non-zero)"; grep -o "only [0-9]* device(s) visible[^;]*" gpurun_out/bench_gpus2.err | head -1
echo "== attention / decode microbench + timelines"; timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu > gpurun_out/attn.log; cat gpurun_out/attn.log
timeout 300 python tools/bench_decode_step.py 4 2>&1 | grep -v "amdgpu\|fused=False" > gpurun_out/decode_step_b4.log
timeout 300 python tools/bench_decode_step.py 1 2>&1 | grep -v "amdgpu\|fused=False" > gpurun_out/decode_step_b1.log; timeout 300 python tools/decode_gemv_timeline.py 4 2>&1 | grep -v amdgpu > gpurun_out/decode_gemv_timeline.log
timeout 300 python tools/conv_layer_profile.py 54 2>&1 | grep -v amdgpu > gpurun_out/conv_layers_54.log
echo "== power probe"; timeout 200 python tools/power_probe.py 2>&1 | grep -v amdgpu > gpurun_out/power_probe.log; cat gpurun_out/power_probe.log
echo "== K-loop anatomy"; ANATOMY=1 timeout 200 python tools/gemm_kloop_anatomy.py 2>&1 | grep -v amdgpu > gpurun_out/kloop_anatomy_final.log; timeout 200 python tools/bench_gemm.py 2>&1 | grep -E "eight-wave|sk=None" > gpurun_out/gemm_microbench.log; cat gpurun_out/gemm_microbench.log
echo "== graph-replay stress"; for sd in 42 7; do timeout 300 python tools/stress_nan.py 300 $sd 1000 2>&1 | grep -v amdgpu | tail -2; done > gpurun_out/stress.log; cat gpurun_out/stress.log
echo "== rocprofv3 kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --eager > $ROOT/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?")
find gpurun_out/prof -type f ! -name "*kernel_stats*" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --eager > $ROOT/gpurun_out/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
  python tools/pmc_summary.py /tmp/pmc_$c gpurun_out/pmc_$c.csv
done
python - <<'PY'
import json
for f in ('bench','bench_strict','bench_split','bench_c4','bench_c5','bench_ddp1'):
    try:
        d=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
        print(f, 'ms/step',round(d['ms_per_step'],3),'value',round(d['value']),d['config'].get('step_launch',''), 'roof', d['roofline'] and {k:d['roofline'][k] for k in ('kernel','achieved','frac','traffic')})
        for k in ('gradient_exchange','artv_train_step','cpu_baseline'):
            if k in d: print('   ',k, d[k])
        for k in d.get('kernels',[]): print('    ',k['kernel'],round(k['ms_per_step'],3),'ms',round(k['tflops'],1),'TF')
    except Exception as e: print(f,'parse',e)
PY
