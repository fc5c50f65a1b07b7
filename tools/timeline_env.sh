#!/bin/bash
# tools/step_timeline.py under several environments (tower order / side-stream switches): the stamps that tell which chain ends the step.
# usage: bash tools/timeline_env.sh <tag> "<ENV_A>" "<ENV_B>" ...   (step_timeline.py under several environments)
tag=$1; shift
mkdir -p gpurun_out/tl
out=gpurun_out/tl/$tag.txt; : > $out
for e in "$@"; do
  echo "== $e" >> $out
  env $e python3 tools/step_timeline.py $TL_ARGS 2>gpurun_out/tl/$tag.err | grep -E "step.start|fwd.end|loss|bwd.start|bwd.end|adam|step.end|gru" >> $out
done
cat $out
