run() { env $1 python bench.py $2 --modes "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  for kv in X=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 HSA_ALLOCATE_QUEUE_DEV_MEM=1 HIP_FORCE_DEV_KERNARG=0; do
    run $kv ""
    run $kv "--config 2"
  done
done
