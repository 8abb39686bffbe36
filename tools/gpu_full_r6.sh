#!/bin/bash
# Round-6 evidence run on the GPU box -> gpurun_out/ : headline bench (CPU baseline + both exact-index legs) FIRST on the fresh box, smoke,
# all GPU tests (one pytest process), exact-mode / config 4 / config 5 / sampling bench lines, the launcher path with the gradient exchange
# forced, kernel timings, rocprofv3 kernel stats of the headline step, FETCH_SIZE / WRITE_SIZE PMC passes (counters only), attention PMC.
# tools/collect_profiles_r6.sh copies the results into profiles/.   usage: bash tools/gpu_full_r6.sh [part ...]   (default: all parts)
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
o=gpurun_out
parts=${*:-"bench tests lines micro census prof pmc"}
python -m mmvid_amd.build > $o/build.log 2>&1
python oracle/build.py >> $o/build.log 2>&1
rocm-smi --showproductname --showmeminfo vram 2>/dev/null | head -20 > $o/rocm_smi.txt
nproc > $o/host.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $o/host.txt
has() { [[ " $parts " == *" $1 "* ]]; }
if has bench; then
  echo "== bench"; timeout 900 python bench.py > $o/bench.log 2> $o/bench.err; echo "bench rc=$?"; grep "bench\]" $o/bench.err | cut -c1-220
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $o/smoke.log | cut -c1-220
fi
if has tests; then
  echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -v -s --timeout 400 -p no:cacheprovider > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $o/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $o/pytest_gpu.log | cut -c1-200
fi
if has lines; then
  for m in split mixed; do echo "== strict $m"; timeout 600 python bench.py --strict $m --steps 20 --no-cpu-baseline --no-exact > $o/bench_$m.log 2> $o/bench_$m.err; grep "bench\]" $o/bench_$m.err | cut -c1-200; done
  echo "== config 4"; timeout 600 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-exact > $o/bench_c4.log 2> $o/bench_c4.err; grep "bench\]" $o/bench_c4.err | cut -c1-200
  for b in 1 4 8 16 32 64; do echo "== config 5, batch $b"; timeout 600 python bench.py --config 5 --batch $b --steps 2 --warmup 1 2>$o/bench_c5_b$b.err | tail -1 > $o/bench_c5_b$b.json; grep "bench\]" $o/bench_c5_b$b.err | cut -c1-200; done
  echo "== BERT sampling (mask-predict)"; timeout 900 python bench.py --sample --steps 3 --warmup 1 > $o/bench_bert_sampling.log 2> $o/bench_bert_sampling.err; grep "bench\]" $o/bench_bert_sampling.err | cut -c1-200
  echo "== launcher, forced exchange"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-exact --force-exchange > $o/bench_ddp1.log 2> $o/bench_ddp1.err; grep "bench\]" $o/bench_ddp1.err | cut -c1-200
fi
if has micro; then
  echo "== kernel timings"; timeout 300 python tools/microbench.py 2>&1 | grep -v amdgpu | tee $o/micro.log
  timeout 300 python tools/bench_gemm_step.py 2>&1 | grep -v amdgpu | tee $o/gemm_step.log
  for m in bf16 mixed split; do timeout 300 python tools/conv_layer_profile.py 54 $m 2>&1 | grep -v amdgpu > $o/conv_layers_54_$m.log; tail -5 $o/conv_layers_54_$m.log | head -2; done
  timeout 300 python tools/bench_decode_step.py 16 2>&1 | grep -v -i "warn\|amdgpu.ids" > $o/decode_step_b16.log; grep -A1 "fused=True" $o/decode_step_b16.log
  timeout 300 python tools/bench_hbm_rows.py 2>&1 | grep -v amdgpu > $o/hbm_rows.log; cat $o/hbm_rows.log | head -12
fi
if has census; then
  echo "== flip census"; timeout 900 python tools/flip_census.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $o/flip_census.log; tail -7 $o/flip_census.log | cut -c1-200
fi
if has prof; then
  echo "== rocprofv3 kernel stats (the headline step alone: eager launches, no other bench legs)"
  rm -rf $o/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$o/prof -o bench -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exact --eager > $ROOT/$o/prof.log 2>&1; echo "rocprof rc=$?")
  find $o/prof -type f ! -name "*kernel_stats*" -delete
  echo "== rocprofv3 kernel stats, config 5 at batch 16 (the decode loop)"
  rm -rf $o/prof_dec
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$o/prof_dec -o dec -- python $ROOT/bench.py --config 5 --batch 16 --steps 1 --warmup 1 > $ROOT/$o/prof_dec.log 2>&1; echo "rocprof rc=$?")
  find $o/prof_dec -type f ! -name "*kernel_stats*" -delete
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact --eager > $ROOT/$o/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
    python tools/pmc_summary.py /tmp/pmc_$c $o/pmc_$c.csv
  done
fi
if has attnpmc; then
  echo "== attention PMC"; bash tools/gpu_pmc_attn.sh 2>&1 | grep -v amdgpu > $o/pmc_attention.txt; grep -E "^==|per MFMA|of wave cycles" $o/pmc_attention.txt
fi
python - <<'PY'
import json
for f in ('bench','bench_split','bench_mixed','bench_c4','bench_ddp1'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/{f}.log') if l.startswith('{')][-1])
        print(f, 'ms/step',round(d['ms_per_step'],3),'value',round(d['value']),d['config'].get('step_launch',''), 'roof', d['roofline'] and {k:d['roofline'][k] for k in ('kernel','achieved','frac','traffic')})
        for k in ('gradient_exchange','cpu_baseline'):
            if k in d: print('   ',k, d[k])
        e=d.get('exact_index_step')
        if e: print('    exact:', {k:e.get(k) for k in ('vae.strict','ms_per_step','ratio_to_headline_step','error')}, 'mixed:', {k:(e.get('near_exact_mixed_operator') or {}).get(k) for k in ('ms_per_step','ratio_to_headline_step')})
        for k in d.get('kernels',[]): print('    ',k['kernel'],round(k['ms_per_step'],3),'ms',round(k['tflops'],1),'TF')
    except Exception as e: print(f,'parse',e)
PY
