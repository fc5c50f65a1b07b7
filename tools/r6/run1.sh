#!/bin/bash
# round 6, GPU call 1: XCD tile blocks + L2 warm-up of conv_halo_rows_kernel (A/B), exactness of the halo tests, in-graph kernel trace
O=gpurun_out/r6a; mkdir -p $O
L="l256.c3x3s1|l512.c3x3s1|l256.c3x3s2|l512.c3x3s2"
for e in "TRICOLO_HALO_XCG=0" "TRICOLO_HALO_TOUCH=0" "TRICOLO_HALO_XCG=-1"; do
  echo "== cold $e" >> $O/layers.txt
  env $e python tools/conv_layers_bench.py --precision f16 --only resnet --layers "$L" --cold --no-wgrad 2>&1 | grep -v amdgpu.ids >> $O/layers.txt
  echo "== hot $e" >> $O/layers.txt
  env $e python tools/conv_layers_bench.py --precision f16 --only resnet --layers "$L" --no-wgrad 2>&1 | grep -v amdgpu.ids >> $O/layers.txt
done
python -m pytest tests -m gpu -x -q -k "halo or bench_plan or conv" > $O/tests_halo.txt 2>&1; tail -3 $O/tests_halo.txt
for e in "TRICOLO_HALO_XCG=0" "TRICOLO_HALO_XCG=-1" "TRICOLO_HALO_XCG=0" "TRICOLO_HALO_XCG=-1"; do
  env $e python bench.py --modes "" --no-cpu-baseline > $O/bench_$e.$RANDOM.json 2>> $O/bench.err
done
python - <<'P'
import glob, json
for f in sorted(glob.glob('gpurun_out/r6a/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f, d['ms_per_step'], r['kernel'][:30], r['frac'], r['frac_raw_events'], r['avg_launch_ms'])
    except Exception as ex:
        print(f, 'ERR', ex)
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --modes "" --no-cpu-baseline --steps 10 --warmup 3 --windows 1 > $GRAFT_REPO_ROOT/$O/trace_bench.json 2> $GRAFT_REPO_ROOT/$O/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $O/trace --shortest > $O/trace_step.txt 2>&1
find $O/trace -name "*.csv" -size +1M -delete
tail -5 $O/trace_step.txt
cat $O/layers.txt
