#!/bin/bash
# Copy the evidence tools/gpu_full_r2.sh left in gpurun_out/ (scratch) into profiles/ (tracked) under this round's names.
cd "$(dirname "$0")/.."
r=${1:-r02}; g=gpurun_out; p=profiles
cp $g/bench.log $p/${r}_bench_n1.json; cp $g/bench.err $p/${r}_bench_n1.stderr.log
cp $g/bench_c4.log $p/${r}_bench_config4.json; cp $g/bench_c5.log $p/${r}_bench_config5.json
cp $g/bench_ddp1.log $p/${r}_bench_launcher_forced_exchange.json
cp $g/smoke.log $p/${r}_smoke.log; cp $g/pytest_gpu.log $p/${r}_gpu_tests.log
cp $g/conv_profile_54.log $p/${r}_vqgan_per_op_profile_54frames.log
cp $g/decode_step_b1.log $p/${r}_artv_decode_step_b1.log; cp $g/decode_step_b4.log $p/${r}_artv_decode_step_b4.log
cp $g/gemm_microbench.log $p/${r}_gemm_microbench.log; cp $g/strip_microbench.log $p/${r}_strip_conv_microbench.log
cp $g/pmc_FETCH_SIZE.csv $p/${r}_pmc_fetch_size.csv; cp $g/pmc_WRITE_SIZE.csv $p/${r}_pmc_write_size.csv
cp $g/prof/bench_kernel_stats.csv $p/${r}_rocprofv3_kernel_stats.csv
cp $g/host.txt $p/${r}_host.txt; cp $g/rocm_smi.txt $p/${r}_rocm_smi.txt
[ -f $g/bench_bert_sampling.log ] && cp $g/bench_bert_sampling.log $p/${r}_bench_bert_sampling.json
[ -f $g/bench_bert_sampling_b3.log ] && cp $g/bench_bert_sampling_b3.log $p/${r}_bench_bert_sampling_3candidates.json
[ -f $g/shape_sweep.log ] && grep -v -E 'amdgpu.ids|Warning|warn' $g/shape_sweep.log > $p/${r}_shape_sweep.log
[ -f $g/step_ops.log ] && cp $g/step_ops.log $p/${r}_framework_launches_per_step.log
[ -f $g/stress_nan.log ] && cp $g/stress_nan.log $p/${r}_graph_replay_stress.log
ls -la $p | grep ${r}_ | wc -l
