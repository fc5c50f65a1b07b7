#!/bin/bash
REPS=10 python tools/r6/determinism.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== old schedule, no touch"; REPS=10 TRICOLO_HALO_TOUCH=0 TRICOLO_HALO_XCG=0 TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0 TRICOLO_PREP_DGRAD_LATE=0 TRICOLO_WGRAD_REDUCE_OVERLAP=0 python tools/r6/determinism.py 2>&1 | grep -v amdgpu.ids | tail -12
