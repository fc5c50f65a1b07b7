#!/bin/bash
# Alternating A/B pairs of bench.py under two environments IN ONE BOX (boxes and processes differ by +-1.5 %: compare within a call only).
# usage: bash tools/ab_env.sh <tag> "<ENV_A>" "<ENV_B>" [bench args...]   (alternating A/B pairs in one box)
tag=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out/ab
out=gpurun_out/ab/$tag.txt; : > $out
for rep in $(seq 1 ${AB_REPS:-3}); do
  for v in A B; do
    if [ $v = A ]; then e="$A"; else e="$B"; fi
    line=$(env $e python3 bench.py --modes "" --no-cpu-baseline "$@" 2>gpurun_out/ab/$tag.err | tail -1)
    echo "$v [$e] $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_windows"]["min"], d["ms_per_step_windows"]["max"], d["value"])')" >> $out
  done
done
cat $out
