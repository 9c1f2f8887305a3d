#!/bin/bash
# same-box: the multi-launch ICP path (--icp-loop 0) with and without the frame's auxiliary records
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['launches_per_frame'])" $1; }
for rep in 1 2; do
  RGBDSLAM_NO_AUX=1 timeout -k 10 200 python bench.py --icp-loop 0 --cpu-frames 0 --no-large-kernels --no-h2d > gpurun_out/auxi_off.json 2> gpurun_out/auxi_off.err || exit 1
  echo "no aux  $(val gpurun_out/auxi_off.json)"
  timeout -k 10 200 python bench.py --icp-loop 0 --cpu-frames 0 --no-large-kernels --no-h2d > gpurun_out/auxi_on.json 2> gpurun_out/auxi_on.err || exit 1
  echo "aux     $(val gpurun_out/auxi_on.json)"
done
