#!/bin/bash
# A/B of library variants on the large kernels of bench.py incl. the mesh / decode of 256 planes: tools/ab_mesh.sh A B ...
for rep in 1 2; do
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 300 python bench.py --cpu-frames 0 --no-h2d --steps 200 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'], [(k['kernel'], k['avg_launch_us'], k['frac']) for k in d['hbm_bound_kernels'][:2]])"
done
done
