#!/bin/bash
# same-box A/B of library variants on the headline bench and the 1080p frame kernels: tools/ab_frame.sh A B ...
for rep in 1 2; do
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 300 python bench.py --cpu-frames 0 --no-h2d > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'], d['roofline']['avg_launch_us'], [(k['kernel'][:22], k['avg_launch_us']) for k in d['hbm_bound_kernels'][2:]])"
done
done
