#!/bin/bash
# PMC passes over one GEMM shape (counters in small groups, each in its own run; kernel-trace only)
cd "$(dirname "$0")/.."
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
which=${1:-fc}
if [ ! -f gpurun_out/pmc/avail.txt ]; then (cd /tmp && timeout 120 rocprofv3 --list-avail > $ROOT/gpurun_out/pmc/avail.txt 2>&1); fi
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA" \
           "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for tile in ${TILES:-128 256}; do
    (cd /tmp && MMVID_GEMM_TILE=$tile timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc/${which}_t${tile}_g$i -o p -- python $ROOT/tools/pmc_gemm.py $which > $ROOT/gpurun_out/pmc/${which}_t${tile}_g$i.log 2>&1)
  done
done
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc/*_g*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'gemm_bf16' in k or 'conv_igemm' in k:
                a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        print(os.path.basename(d), {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
find gpurun_out/pmc -type f -name "*.csv" ! -name "*counter_collection*" -delete
