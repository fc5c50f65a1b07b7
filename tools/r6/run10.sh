#!/bin/bash
O=gpurun_out/r6j; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q -k "halo or bench_plan or conv_16bit or adam" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for e in "TRICOLO_HALO_TN2=1" "TRICOLO_HALO_TN2=0"; do
  echo "== $e hot"; env $e python tools/conv_layers_bench.py --precision f16 --only resnet --layers "l512.c3x3s1" --no-wgrad 2>&1 | grep -v amdgpu.ids
  echo "== $e cold"; env $e python tools/conv_layers_bench.py --precision f16 --only resnet --layers "l512.c3x3s1" --no-wgrad --cold 2>&1 | grep -v amdgpu.ids
done
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do
  run "TRICOLO_HALO_TN2=1" tn2 $rep
  run "TRICOLO_HALO_TN2=0" tn4 $rep
  run "TRICOLO_ADAM_KARG=0" nokarg $rep
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6j/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], r['kernel'][:24], r['frac'], r['avg_launch_ms'], r['families']['conv_halo_rows_kernel']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    for x in v: print(k, x)
P
tail -3 $O/bench.err
python tools/step_timeline.py 2>/dev/null | grep -E "step.start|fwd.end|loss|bwd.start|bwd.end|adam|step.end"
