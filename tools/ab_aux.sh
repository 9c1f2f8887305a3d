#!/bin/bash
# same-box A/B of the frame's auxiliary ICP records (RGBDSLAM_NO_AUX=1: the ICP forms the four values per sample itself)
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d.get('stage_times',{}).get('stage4_fusion'))" $1; }
for rep in 1 2 3; do
  RGBDSLAM_NO_AUX=1 timeout -k 10 200 python bench.py --cpu-frames 0 --no-large-kernels --no-h2d > gpurun_out/aux_off.json 2> gpurun_out/aux_off.err || exit 1
  echo "no aux  $(val gpurun_out/aux_off.json)"
  timeout -k 10 200 python bench.py --cpu-frames 0 --no-large-kernels --no-h2d > gpurun_out/aux_on.json 2> gpurun_out/aux_on.err || exit 1
  echo "aux     $(val gpurun_out/aux_on.json)"
done
