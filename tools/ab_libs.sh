#!/bin/bash
# A/B of two BUILDS in one gpurun call (boxes differ by +-1.5 %): put the other build at tricolo_amd/libtricolo_hip_base.so (git-ignored,
# travels with the snapshot); three alternating bench.py runs each.  AB_ARGS="--config 3" etc.
cd $GRAFT_REPO_ROOT
cp tricolo_amd/libtricolo_hip.so /tmp/new.so
for i in 1 2 3; do
  for which in new base; do
    if [ $which = base ]; then cp tricolo_amd/libtricolo_hip_base.so tricolo_amd/libtricolo_hip.so; else cp /tmp/new.so tricolo_amd/libtricolo_hip.so; fi
    python bench.py $AB_ARGS --modes "" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', b['ms_per_step'], b['ms_per_step_windows']['min'])"
  done
done
cp /tmp/new.so tricolo_amd/libtricolo_hip.so
