#!/bin/bash
# same-box A/B of the end-of-sequence path at the driver's own arguments (--steps 20 --warmup 5)
set -e
ARGS="--steps 20 --warmup 5 --cpu-frames 0 --no-large-kernels --no-h2d"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" $1; }
for i in 1 2 3; do
  RGBDSLAM_SYNC_SPIN_US=0 RGBDSLAM_SYNC_COPY_CMDS=1 RGBDSLAM_SEQ_LEADING_SYNC=1 timeout -k 10 100 python bench.py $ARGS > gpurun_out/ab_old.json 2> gpurun_out/ab_old.err
  echo "old      $(val gpurun_out/ab_old.json)"
  timeout -k 10 100 python bench.py $ARGS > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err
  echo "new      $(val gpurun_out/ab_new.json)"
  RGBDSLAM_SYNC_SPIN_US=0 timeout -k 10 100 python bench.py $ARGS > gpurun_out/ab_x.json 2> gpurun_out/ab_x.err
  echo "new-spin $(val gpurun_out/ab_x.json)"
  RGBDSLAM_SYNC_COPY_CMDS=1 timeout -k 10 100 python bench.py $ARGS > gpurun_out/ab_x.json 2> gpurun_out/ab_x.err
  echo "new-copy $(val gpurun_out/ab_x.json)"
done
RGBDSLAM_SEQ_TIMING=1 RGBDSLAM_SYNC_SPIN_US=0 RGBDSLAM_SYNC_COPY_CMDS=1 RGBDSLAM_SEQ_LEADING_SYNC=1 timeout -k 10 100 python bench.py $ARGS 2>&1 >/dev/null | grep "seq timing" | tail -3
echo ---
RGBDSLAM_SEQ_TIMING=1 timeout -k 10 100 python bench.py $ARGS 2>&1 >/dev/null | grep "seq timing" | tail -3
