#!/bin/bash
# A/B of library variants on the config-3 kernels (decode and mesh of 256 planes x 512^2): tools/ab_decode.sh <variant> ...
cp rgb-d-slam_amd/librgbdslam_hip.so /tmp/librgbdslam_hip.keep || exit 1
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 300 python bench.py --cpu-frames 0 --steps 50 --warmup 5 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', [(k['kernel'], k['avg_launch_us'], k['achieved']) for k in d['hbm_bound_kernels'] if 'bump' in k['kernel']])"
done
cp /tmp/librgbdslam_hip.keep rgb-d-slam_amd/librgbdslam_hip.so
