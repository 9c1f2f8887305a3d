#!/bin/bash
# A/B of library variants on the large-size kernel rooflines of bench.py (1080p fused frame kernels, 256-plane decode):
#   tools/ab_large.sh A B ...   (variants/<name>/librgbdslam_hip.so, see tools/make_variant.sh)
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 300 python bench.py --cpu-frames 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'], d['roofline']['avg_launch_us'], [(k['kernel'], k['avg_launch_us'], k['frac']) for k in d['hbm_bound_kernels'][2:]])"
done
