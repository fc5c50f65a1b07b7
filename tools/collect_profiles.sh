#!/bin/bash
# Copy what tools/profile_round.sh <tag> merged back into gpurun_out/ into the tracked profiles/<tag>/ (the judge reads profiles/).
tag=${1:-r5}
O=gpurun_out
P=profiles/$tag
mkdir -p $P
cp $O/${tag}_bench_default.json $P/bench_default.json
for c in 2 3 5; do cp $O/${tag}_bench_cfg$c.json $P/bench_config$c.json; done
for c in 2 3 4 5; do cp $O/${tag}_timeline_cfg$c.txt $P/timeline_config$c.txt; done
for p in f16 bf16x3; do
  # (gpurun merges every call's files into gpurun_out/: take the newest run's stats, not the first name found)
  f=$(find $O/${tag}_prof_$p -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $P/kernel_stats_$p.csv
  cp $O/${tag}_prof_$p.json $P/bench_under_rocprof_$p.json
done
for c in cfg2 cfg5 voxel_fwd; do
  f=$(find $O/${tag}_prof_$c -name "*kernel_stats.csv" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $P/kernel_stats_$c.csv
done
cp $O/${tag}_pmc_traffic_f16.json $P/pmc_traffic_f16.json
cp $O/${tag}_pmc_mfma_f16.json $P/pmc_mfma_f16.json
cp $O/${tag}_voxel_fwd.txt $P/voxel_fwd.txt
cp $O/${tag}_kernel_times.txt $P/kernel_times.txt
for p in f16 bf16; do cp $O/${tag}_conv_layers_$p.txt $P/conv_layers_$p.txt; done
cp $O/${tag}_gpu_tests.txt $P/gpu_tests.txt
cp $O/parity_report.json $P/parity_report.json
ls -la $P
