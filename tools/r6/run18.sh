#!/bin/bash
REPS=3000 N=1 python tools/r6/determinism.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300
OLD="TRICOLO_HALO_TOUCH=0 TRICOLO_HALO_XCG=0 TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0 TRICOLO_PREP_DGRAD_LATE=0 TRICOLO_WGRAD_REDUCE_OVERLAP=0"
env $OLD TRICOLO_DS_FWD=2 REPS=1500 N=1 python tools/r6/determinism.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
python tools/kernel_times.py 2>/dev/null | grep -E "gru|entry-point"
