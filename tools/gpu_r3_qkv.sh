#!/bin/bash
# fused q|k|v 1x1 conv of the VQGAN AttnBlocks: parity tests, then interleaved bench A/B (separate processes: env read at import)
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "vqgan or vae or encode or config2 or golden" 2>&1 | tail -5 > gpurun_out/qkv_tests.log
for i in 1 2 3; do
  for f in 0 1; do
    echo "fuse_qkv=$f" >> gpurun_out/qkv_ab.log
    MMVID_FUSE_QKV=$f timeout 300 python bench.py --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/qkv_ab.log
  done
done
cat gpurun_out/qkv_tests.log gpurun_out/qkv_ab.log
