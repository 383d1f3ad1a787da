#!/bin/bash
# copy one run_profiles.sh result set (gpurun_out/<tag>) into profiles/ under the round's names: tools/copy_profiles.sh <tag> <rNN>
S=gpurun_out/$1; R=$2
cp $S/bench_default.json profiles/${R}_bench_bf16_default.json
cp $S/bench_wgrad0.json profiles/${R}_bench_bf16_site_wgrad0.json
cp $S/bench_ensemble24.json profiles/${R}_bench_ensemble24.json
cp $S/bench_ensemble24_wgrad0.json profiles/${R}_bench_ensemble24_site_wgrad0.json
cp $S/dp_one_rank.txt profiles/${R}_dp_one_rank.txt
cp $S/batch_sweep.txt profiles/${R}_batch_sweep.txt
for f in multistage multistage_predict probe_stream probe_events traffic_ensemble24; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/${R}_$f.txt; done
[ -f $S/traffic_ensemble24.json ] && cp $S/traffic_ensemble24.json profiles/${R}_traffic_ensemble24.json
cp $S/private_segment.txt profiles/${R}_private_segment.txt
cp $S/infer.txt profiles/${R}_inference.txt
cp $S/kernel_trace.txt profiles/${R}_kernel_trace_bf16_B1024.txt
cp $S/kernel_trace_ensemble24.txt profiles/${R}_kernel_trace_ensemble24.txt
[ -f $S/kernel_trace_multistage_B128.txt ] && cp $S/kernel_trace_multistage_B128.txt profiles/${R}_kernel_trace_multistage_B128.txt
cp $S/kernel_trace_dispatches.csv profiles/${R}_kernel_trace_bf16_B1024_dispatches.csv
cp $S/kernel_trace_ensemble24_dispatches.csv profiles/${R}_kernel_trace_ensemble24_dispatches.csv
cp $S/pmc_fetch.csv profiles/${R}_pmc_fetch.csv
cp $S/pmc_write.csv profiles/${R}_pmc_write.csv
cp $S/pmc_sq.csv profiles/${R}_pmc_sq.csv
cp $S/traffic_step.json profiles/${R}_traffic_step.json
[ -f $S/gpu_tests.txt ] && cp $S/gpu_tests.txt profiles/${R}_gpu_tests.txt
