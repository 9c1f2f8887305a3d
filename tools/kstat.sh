#!/bin/bash
# usage: kstat.sh variant  -> rocprofv3 kernel stats of a short bench run with that variant
v=$1
cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$v -o k -- python3 /root/repo/bench.py --cpu-frames 0 --no-large-kernels --steps 200 --warmup 20 > /dev/null 2> /root/repo/gpurun_out/prof_$v.err || { tail -3 /root/repo/gpurun_out/prof_$v.err; exit 1; }
db=$(ls /root/repo/gpurun_out/prof_$v/*/k_results.db /root/repo/gpurun_out/prof_$v/k_results.db 2>/dev/null | head -1)
python3 /root/repo/tools/rocpd_summary.py $db | head -9
rm -rf /root/repo/gpurun_out/prof_$v
