#!/bin/bash
# Round evidence recipe (run on the GPU box from the repo root):  bash tools/profile_round.sh <tag>
# Writes gpurun_out/<tag>_*: the -m gpu test log, bench.py JSON lines (default run + configs 2/3/5), rocprofv3 kernel stats
# for the f16 and bf16x3 modes, the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, MI355X_MICROARCH.md) and the
# in-graph step timelines (tools/step_timeline.py).
tag=${1:-r6}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q > $O/${tag}_gpu_tests_full.txt 2>&1; grep -E "passed|failed|error" $O/${tag}_gpu_tests_full.txt | tail -5 > $O/${tag}_gpu_tests.txt
python bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
for c in 2 3 5; do python bench.py --config $c --modes "" --no-cpu-baseline > $O/${tag}_bench_cfg$c.json 2> $O/${tag}_bench_cfg$c.err; done
for c in 4 2 3 5; do python tools/step_timeline.py --config $c > $O/${tag}_timeline_cfg$c.txt 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
for p in f16 bf16x3; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_$p -- python3 $R/bench.py --steps 20 --warmup 5 --precision $p --modes "" --no-cpu-baseline > $O/${tag}_prof_$p.json 2> $O/${tag}_prof_$p.err
done
# (round 5) kernel stats of the other bench configurations and of a voxel-forward-only run (the five SubMConv3d launches)
for c in 2 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_cfg$c -- python3 $R/bench.py --config $c --steps 20 --warmup 5 --precision f16 --modes "" --no-cpu-baseline > $O/${tag}_prof_cfg$c.json 2> $O/${tag}_prof_cfg$c.err
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_voxel_fwd -- python3 $R/tools/voxel_fwd_bench.py --modes f16 > $O/${tag}_prof_voxel_fwd.txt 2> $O/${tag}_prof_voxel_fwd.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --precision f16 --modes "" --no-cpu-baseline > $O/${tag}_pmc_fetch.json 2> $O/${tag}_pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_write -- python3 $R/bench.py --steps 3 --warmup 2 --precision f16 --modes "" --no-cpu-baseline > $O/${tag}_pmc_write.json 2> $O/${tag}_pmc_write.err
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${tag}_pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 2 --precision f16 --modes "" --no-cpu-baseline > /dev/null 2> $O/${tag}_pmc_mfma.err
cd $R
python tools/pmc_mfma.py $O/${tag}_pmc_mfma > $O/${tag}_pmc_mfma_f16.json
python tools/pmc_traffic.py $O/${tag}_pmc_fetch $O/${tag}_pmc_write f16 > $O/${tag}_pmc_traffic_f16.json
python tools/kernel_times.py > $O/${tag}_kernel_times.txt 2>/dev/null
python tools/voxel_fwd_bench.py --modes f16,bf16 --out $O/${tag}_voxel_fwd.txt --json $O/${tag}_voxel_fwd.json > /dev/null 2>&1
for p in f16 bf16; do python tools/conv_layers_bench.py --precision $p 2>&1 | grep -v amdgpu.ids > $O/${tag}_conv_layers_$p.txt; done
# keep the merged-back payload small: kernel traces are large, the stats CSVs are what gets committed
find $O/${tag}_prof_* -name "*kernel_trace.csv" -size +20M -delete
ls -la $O | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}_smoke.txt 2>&1; tail -3 $O/${tag}_smoke.txt
