import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["cells_fused"], d["predicted_coverage"], {k:v for k,v in d["phases"].items() if k!="unit"})
        for k in d["planes_kernels"]: print("   ", k["kernel"], k["avg_launch_us"], k["frac"])
    except Exception as e: print(f, "ERR", e)
