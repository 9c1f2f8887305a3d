#!/bin/bash
# A/B of library variants on one box: tools/ab.sh "<bench args>" A B ...
# variants/<name>/librgbdslam_hip.so are builds of other revisions / flags made by tools/make_variant.sh NAME [GIT_REV]
# (git-ignored scratch, shipped to the box by gpurun; the working library is restored at the end).
args="$1"; shift
for rep in 1 2; do
for v in "$@"; do
  cp variants/$v/librgbdslam_hip.so rgb-d-slam_amd/librgbdslam_hip.so || exit 1
  timeout -k 10 200 python bench.py --no-large-kernels --cpu-frames 0 $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
print(sys.argv[1], d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_per_frame"])
PY
done
done
