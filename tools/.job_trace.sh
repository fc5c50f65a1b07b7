cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4t
cd $R
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4t/c3 -- python3 bench.py --config 3 --per-gpu-batch 32 --modes "" --no-cpu-baseline --steps 10 --warmup 3 --windows 1 --preroll 5 > gpurun_out/r4t/c3.json 2> gpurun_out/r4t/c3.err
python3 tools/trace_step.py gpurun_out/r4t/c3 --shortest > gpurun_out/r4t/trace_c3_b32.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4t/c4 -- python3 bench.py --modes "" --no-cpu-baseline --steps 10 --warmup 3 --windows 1 --preroll 5 > gpurun_out/r4t/c4.json 2> gpurun_out/r4t/c4.err
python3 tools/trace_step.py gpurun_out/r4t/c4 --shortest > gpurun_out/r4t/trace_c4.txt 2>&1
rm -rf gpurun_out/r4t/c3 gpurun_out/r4t/c4
tail -3 gpurun_out/r4t/trace_c3_b32.txt; tail -3 gpurun_out/r4t/trace_c4.txt
