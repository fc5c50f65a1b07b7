#!/bin/bash
# round 6, GPU call 7: late packing of the data-gradient operands + input-tile warm-up in conv_halo_rows_kernel (A/B), halo exactness tests
O=gpurun_out/r6g; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "halo or bench_plan or training_steps or f16_mode" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
run() { env $1 python bench.py --modes "" --no-cpu-baseline > $O/bench_$2.$3.json 2>> $O/bench.err; }
for rep in 1 2 3; do
  run "TRICOLO_PREP_DGRAD_LATE=1" late1 $rep
  run "TRICOLO_PREP_DGRAD_LATE=0" late0 $rep
  run "TRICOLO_HALO_TOUCH=0" notouch $rep
done
python - <<'P'
import glob, json, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r6g/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        res[f.split('/')[-1].split('.')[0]].append((d['ms_per_step'], r['kernel'][:22], r['frac'], r['avg_launch_ms']))
    except Exception as ex:
        print(f, 'ERR', ex)
for k, v in sorted(res.items()):
    print(k, v)
P
python tools/conv_layers_bench.py --precision f16 --only resnet --layers "l256.c3x3s1|l512.c3x3s1" --cold --no-wgrad 2>&1 | grep -v amdgpu.ids
