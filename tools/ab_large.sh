#!/bin/bash
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 300 python bench.py --cpu-frames 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'], [(k['kernel'], k['avg_launch_us']) for k in d['hbm_bound_kernels'] if 'merge' in k['kernel']])"
done
