#!/bin/bash
# same-box look at what the driver's --steps 20 --warmup 5 run pays against longer runs
set -e
COMMON="--cpu-frames 0 --no-large-kernels --no-h2d"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['icp_iterations_per_level'])" $1; }
run() { timeout -k 10 100 python bench.py $COMMON "$@" > gpurun_out/k20.json 2> gpurun_out/k20.err; echo "$* -> $(val gpurun_out/k20.json)"; }
for i in 1 2; do
  run --steps 20 --warmup 5
  run --steps 20 --warmup 5 --preheat-ms 0
  run --steps 20 --warmup 60
  run --steps 40 --warmup 5
  run --steps 200 --warmup 5
  run --steps 2200 --warmup 5
done
