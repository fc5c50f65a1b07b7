R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python3 $R/bench.py --config 2 --steps 20 --warmup 5 --modes "" --no-cpu-baseline > /dev/null 2>&1
cd $R; find gpurun_out/prof_cfg2 -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_cfg2/*/*kernel_stats.csv")[0])))
n = [int(r["Calls"]) for r in rows if r["Name"].startswith("adam_seg")][0]
for r in rows[:30]:
    print(f'{r["Name"][:80]:82s} {int(r["Calls"])/n:6.1f} avg {float(r["AverageNs"])/1e3:9.1f} us  tot/step {float(r["TotalDurationNs"])/n/1e3:8.1f}')
PY
python tools/step_timeline.py --config 2 | tail -12
