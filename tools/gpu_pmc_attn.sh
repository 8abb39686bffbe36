#!/bin/bash
# PMC passes over the attention kernels (counters in small groups, each in its own run; kernel-trace only)
cd "$(dirname "$0")/.."
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc/attn_g$i -o p -- python $ROOT/tools/microbench.py attention > $ROOT/gpurun_out/pmc/attn_g$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, os
for kern in ('attn_fwd', 'attn_bwd_dq', 'attn_bwd_dkv'):
    tot = {}
    for d in sorted(glob.glob('gpurun_out/pmc/attn_g*')):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                if kern in r['Kernel_Name']:
                    a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
            for k, v in acc.items(): tot[k] = v[0] / max(v[1], 1)
    print('==', kern)
    for k in sorted(tot): print(f'   {k:28s} {tot[k]:16.0f}')
    m = tot.get('SQ_INSTS_MFMA', 0); wc = tot.get('SQ_WAVE_CYCLES', 1)
    if m:
        print('   per MFMA: VALU %.2f SALU %.2f LDS %.2f VMEM %.3f' % (tot.get('SQ_INSTS_VALU',0)/m, tot.get('SQ_INSTS_SALU',0)/m, tot.get('SQ_INSTS_LDS',0)/m, tot.get('SQ_INSTS_VMEM',0)/m))
        print('   of wave cycles: WAIT_ANY %.2f WAIT_INST_ANY %.2f ACTIVE_INST_ANY %.2f (VALU %.2f LDS %.2f) ; MFMA busy / (4 x wave quad-cycles) %.3f' % (
            tot.get('SQ_WAIT_ANY',0)/wc, tot.get('SQ_WAIT_INST_ANY',0)/wc, tot.get('SQ_ACTIVE_INST_ANY',0)/wc, tot.get('SQ_ACTIVE_INST_VALU',0)/wc, tot.get('SQ_ACTIVE_INST_LDS',0)/wc, tot.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*wc)))
        print('   LDS bank conflict / LDS active %.3f' % (tot.get('SQ_LDS_BANK_CONFLICT',0)/max(tot.get('SQ_LDS_IDX_ACTIVE',1),1)))
PY
find gpurun_out/pmc -type f -name "*.csv" -delete
