#!/bin/bash
# tools/kernel_times.py (unprofiled per-entry-point times of eager steps) under several environments, filtered by a pattern.
# usage: bash tools/kernel_times_env.sh <tag> <grep pattern> "<ENV_A>" "<ENV_B>" ...   (kernel_times.py under several environments)
tag=$1; pat=$2; shift 2
mkdir -p gpurun_out/kt
out=gpurun_out/kt/$tag.txt; : > $out
for e in "$@"; do
  echo "== $e" >> $out
  env $e python3 tools/kernel_times.py --modes "" --no-cpu-baseline 2>gpurun_out/kt/$tag.err | grep -E "$pat|entry-point" >> $out
done
cat $out
