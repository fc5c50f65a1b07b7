#!/bin/bash
O=gpurun_out/r6ag; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --modes "" --no-cpu-baseline --no-voxel-config5 --steps 10 --warmup 3 --windows 1 > $GRAFT_REPO_ROOT/$O/trace_bench.json 2> $GRAFT_REPO_ROOT/$O/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $O/trace --shortest > $O/trace_step.txt 2>&1
find $O/trace -name "*.csv" -size +1M -delete
wc -l $O/trace_step.txt
