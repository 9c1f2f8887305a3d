#!/usr/bin/env python3
"""profiles/<tag>_pmc_fetch_size.txt + <tag>_pmc_write_size.txt (tools/rocpd_pmc.py dumps) -> <tag>_pmc_traffic.json,
the file bench.py reads `roofline.traffic` from.  usage: tools/pmc_traffic_json.py profiles/r01_g"""
import json, re, sys
base = sys.argv[1]
out = {}
for kind, key in (("fetch_size", "fetch"), ("write_size", "write")):
    for line in open("%s_pmc_%s.txt" % (base, kind)):
        # the kernel name may hold blanks ("k_icp_loop<false, false>"): everything in front of the counter's name
        mt = re.match(r"^(k_.*?)\s+([A-Z][A-Z_0-9]+)\s+(\d+)\s+(\d+)\s+([\d.]+)", line)
        if not mt:
            continue
        name, calls, avg = mt.group(1), int(mt.group(4)), float(mt.group(5))
        e = out.setdefault(name, {"fetch": 0, "write": 0, "launches": 0})
        # several grid sizes of one kernel: launch-weighted mean
        tot = e[key] * e.get("_n_" + key, 0) + avg * 1024.0 * calls
        e["_n_" + key] = e.get("_n_" + key, 0) + calls
        e[key] = tot / e["_n_" + key]
        e["launches"] = max(e["launches"], e["_n_" + key])
for e in out.values():
    for k in list(e):
        if k.startswith("_n_"):
            del e[k]
    e["fetch"], e["write"] = int(round(e["fetch"])), int(round(e["write"]))
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh) on: bench.py --cpu-frames 0 "
                 "--steps 40 --warmup 5 --no-large-kernels --no-h2d; unit of the counters: KB; raw values, no gfx950 x2 FETCH correction "
                 "applied (the accesses are 4-16 B per lane, uncalibrated per MI355X_MICROARCH.md)",
       "bytes_per_launch": out}
json.dump(doc, open(base + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("k_icp_")}))
