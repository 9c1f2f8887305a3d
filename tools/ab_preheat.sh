#!/bin/bash
# same-box: the driver's --steps 20 --warmup 5 against the length of the untimed pre-heat
set -e
COMMON="--cpu-frames 0 --no-large-kernels --no-h2d --steps 20 --warmup 5"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" $1; }
run() { timeout -k 10 100 python bench.py $COMMON "$@" > gpurun_out/ph.json 2> gpurun_out/ph.err; echo "$* -> $(val gpurun_out/ph.json)"; }
for i in 1 2 3; do
  run --preheat-ms 120
  run --preheat-ms 400
  run --preheat-ms 1200
done
