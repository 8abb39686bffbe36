#!/bin/bash
# Copy the evidence tools/gpu_full_r6.sh left in gpurun_out/ (scratch) into profiles/ (tracked) under this round's names.
cd "$(dirname "$0")/.."
r=r06; g=gpurun_out; p=profiles
c() { [ -s "$1" ] && cp "$1" "$2"; }
c $g/bench.log $p/${r}_bench_n1.json; c $g/bench.err $p/${r}_bench_n1.stderr.log
c $g/bench_split.log $p/${r}_bench_strict_split.json; c $g/bench_mixed.log $p/${r}_bench_strict_mixed.json
c $g/bench_c4.log $p/${r}_bench_config4.json
for b in 1 4 8 16 32 64; do c $g/bench_c5_b$b.json $p/${r}_bench_config5_b$b.json; done
c $g/bench_ddp1.log $p/${r}_bench_launcher_forced_exchange.json
c $g/bench_bert_sampling.log $p/${r}_bench_bert_sampling.json
c $g/smoke.log $p/${r}_smoke.log; c $g/pytest_gpu.log $p/${r}_gpu_tests.log
c $g/micro.log $p/${r}_kernel_timings.log; c $g/gemm_step.log $p/${r}_gemm_layer_calls.log
for m in bf16 mixed split; do c $g/conv_layers_54_$m.log $p/${r}_vqgan_encoder_per_layer_54_frames_$m.log; done
c $g/hbm_rows.log $p/${r}_hbm_rows_microbench.log; c $g/decode_step_b16.log $p/${r}_decode_step_b16.log
c $g/flip_census.log $p/${r}_flip_census.log
c $g/pmc_FETCH_SIZE.csv $p/${r}_pmc_fetch_size.csv; c $g/pmc_WRITE_SIZE.csv $p/${r}_pmc_write_size.csv
c $g/prof/bench_kernel_stats.csv $p/${r}_rocprofv3_kernel_stats.csv
c $g/prof_dec/dec_kernel_stats.csv $p/${r}_rocprofv3_config5_b16_kernel_stats.csv
c $g/host.txt $p/${r}_host.txt; c $g/rocm_smi.txt $p/${r}_rocm_smi.txt
ls -la $p | grep ${r}_ | wc -l
