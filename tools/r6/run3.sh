#!/bin/bash
# round 6, GPU call 3: issue-order A/B of the side branches (shortcut fwd / bwd, trunk packing), toolchain probe test, cold / hot layer table
O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "toolchain_probe" > $O/tests_probe.txt 2>&1; tail -5 $O/tests_probe.txt
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2; do
  run "TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0" base $rep
  run "TRICOLO_DS_FWD=1 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0" fwd1 $rep
  run "TRICOLO_DS_FWD=2 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=0" fwd2 $rep
  run "TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=1 TRICOLO_PREP_ISSUE=0" bwd1 $rep
  run "TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=2 TRICOLO_PREP_ISSUE=0" bwd2 $rep
  run "TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=1" prep1 $rep
  run "TRICOLO_DS_FWD=0 TRICOLO_DS_BWD=0 TRICOLO_PREP_ISSUE=2" prep2 $rep
  run "TRICOLO_DS_FWD=2 TRICOLO_DS_BWD=2 TRICOLO_PREP_ISSUE=1" all $rep
done
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/r6c/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d['ms_per_step_windows']['min'], d['config']['final_loss'])
    except Exception as ex:
        print(f, 'ERR', ex)
P
python tools/conv_layers_bench.py --precision f16 --cold > $O/layers_cold.txt 2>&1
python tools/conv_layers_bench.py --precision f16 > $O/layers_hot.txt 2>&1
grep -v amdgpu.ids $O/layers_cold.txt; grep -v amdgpu.ids $O/layers_hot.txt
