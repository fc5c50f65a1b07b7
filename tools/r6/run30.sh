#!/bin/bash
O=gpurun_out/r6ab; mkdir -p $O; rm -f $O/*
TRICOLO_FINE_STAMPS=1 timeout 300 python tools/step_timeline.py > $O/fine.txt 2>/dev/null
wc -l $O/fine.txt
