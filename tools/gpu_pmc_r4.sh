cd /root/repo; ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_pmc_attn.sh 2>&1 | grep -v amdgpu > gpurun_out/pmc_attention.txt
rm -rf /tmp/pmc_cal; (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_cal -o p -- python $ROOT/tools/fetch_calibration.py > $ROOT/gpurun_out/pmc_cal.log 2>&1)
python tools/pmc_summary.py /tmp/pmc_cal gpurun_out/pmc_fetch_calibration.csv
rm -rf /tmp/pmc_calw; (cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_calw -o p -- python $ROOT/tools/fetch_calibration.py > $ROOT/gpurun_out/pmc_calw.log 2>&1)
python tools/pmc_summary.py /tmp/pmc_calw gpurun_out/pmc_write_calibration.csv
cat gpurun_out/pmc_attention.txt; cat gpurun_out/pmc_fetch_calibration.csv gpurun_out/pmc_write_calibration.csv
