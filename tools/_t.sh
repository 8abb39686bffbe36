cd /root/repo
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemv or decode or artv or kv_cache" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -3
bash tools/gpu_ab_lib.sh base "for b in 8 16; do python bench.py --config 5 --batch \$b --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('batch', d['config']['per_gpu_batch'], round(d['value']), 'sampled tokens/s', round(d['roofline']['ms_per_token_step']*1e3,1), 'us per token')\"; done" 2
