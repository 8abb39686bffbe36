#!/bin/bash
# Copy the evidence tools/gpu_full_r4.sh left in gpurun_out/ (scratch) into profiles/ (tracked) under this round's names.
cd "$(dirname "$0")/.."
r=r04; g=gpurun_out; p=profiles
cp $g/bench.log $p/${r}_bench_n1.json; cp $g/bench.err $p/${r}_bench_n1.stderr.log
cp $g/bench_strict.log $p/${r}_bench_strict.json; cp $g/bench_split.log $p/${r}_bench_strict_split.json
cp $g/bench_c4.log $p/${r}_bench_config4.json; cp $g/bench_c4_b20.log $p/${r}_bench_config4_batch20.json; cp $g/bench_c5.log $p/${r}_bench_config5.json
cp $g/bench_ddp1.log $p/${r}_bench_launcher_forced_exchange.json
cp $g/power_probe.log $p/${r}_power_probe_mfma_ceiling.log; cp $g/kloop_anatomy_final.log $p/${r}_gemm_kloop_anatomy.log; cp $g/gemm_microbench.log $p/${r}_gemm_microbench.log
cp $g/bench_bert_sampling.log $p/${r}_bench_bert_sampling.json
cp $g/smoke.log $p/${r}_smoke.log; cp $g/pytest_gpu.log $p/${r}_gpu_tests.log
cp $g/attn.log $p/${r}_attention_microbench.log
cp $g/decode_step_b4.log $p/${r}_artv_decode_step_b4.log; cp $g/decode_step_b1.log $p/${r}_artv_decode_step_b1.log; cp $g/decode_gemv_timeline.log $p/${r}_decode_gemv_timeline.log
cp $g/conv_layers_54.log $p/${r}_vqgan_encoder_per_layer_54_frames.log
cp $g/pmc_FETCH_SIZE.csv $p/${r}_pmc_fetch_size.csv; cp $g/pmc_WRITE_SIZE.csv $p/${r}_pmc_write_size.csv
cp $g/prof/bench_kernel_stats.csv $p/${r}_rocprofv3_kernel_stats.csv
cp $g/stress.log $p/${r}_graph_replay_stress.log
cp $g/host.txt $p/${r}_host.txt; cp $g/rocm_smi.txt $p/${r}_rocm_smi.txt
grep -o "only [0-9]* device(s) visible[^;]*" $g/bench_gpus2.err | head -1 > $p/${r}_bench_gpus2_on_one_gpu_box.txt
ls -la $p | grep ${r}_ | wc -l
