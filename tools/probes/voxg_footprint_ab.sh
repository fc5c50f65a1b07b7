#!/bin/bash
# Does a SMALLER footprint of the voxel tower's coarse-grid kernels (fewer, fatter workgroups: channel tile 64) shorten the TRIMODAL step even though
# the kernels themselves get slower?  (Image-tower kernels are gangs of ~one workgroup per CU: every CU a side tower holds delays one of them.)
for rep in 1 2 3; do
  python bench.py --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default      ', d['ms_per_step'])"
  TRICOLO_VOXG_CT=64 python bench.py --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('voxg ct 64   ', d['ms_per_step'])"
  TRICOLO_NO_VOXG=1 python bench.py --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no voxg (r4) ', d['ms_per_step'])"
done
