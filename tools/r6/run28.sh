#!/bin/bash
O=gpurun_out/r6z; mkdir -p $O; rm -f $O/*
for v in new nokernel nocover; do
TRICOLO_GUARD_DBG=$v timeout 300 python tools/step_timeline.py 2>/dev/null | grep -E "image.bwd.end|adam|step.end" | sed "s/^/$v /"
done
TRICOLO_GUARD_NOTE=0 timeout 300 python tools/step_timeline.py 2>/dev/null | grep -E "image.bwd.end|adam|step.end" | sed "s/^/old /"
