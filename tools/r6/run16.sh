#!/bin/bash
O=gpurun_out/r6q; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; grep -E "passed|failed|error" $O/tests.txt | tail -3
python tools/step_timeline.py 2>/dev/null | grep -E "step.start|stem|layer|fwd.end|loss|bwd|adam|step.end"
