#!/bin/bash
# like tools/ab.sh, but prints the rocprofv3 per-kernel averages of each variant: tools/ab_kernels.sh A B
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 200 python bench.py --no-large-kernels --cpu-frames 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['value'])"
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/abk_$v -o k -- python3 /root/repo/bench.py --cpu-frames 0 --no-large-kernels > /dev/null 2>&1) || exit 1
  db=$(ls gpurun_out/abk_$v/*/k_results.db gpurun_out/abk_$v/k_results.db 2>/dev/null | head -1)
  python3 tools/rocpd_summary.py $db | head -9
  rm -rf gpurun_out/abk_$v
done
