"""The bench line's contract, checked on the line committed with the round's artefacts (profiles/*_bench.json is what
`python bench.py` printed on the GPU box): the keys the driver reads, the roofline and CPU-baseline objects."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line"
    text = open(files[-1]).read().strip()
    assert "\n" not in text, "bench.py prints ONE line"
    return json.loads(text)


def test_top_level_keys_and_types():
    d = latest_line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] in base["metric"]                      # "ICP+fusion frames/sec at 640x480" is BASELINE's metric
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == 1 and d["steps"] > 0 and d["warmup"] >= 0
    assert abs(d["value"] - 1e3 * d["n_gpus"] / d["ms_per_step"]) < 0.01 * d["value"]   # whole-job frames/s
    assert d["vs_baseline"] is None and base["published"] == {}                       # no published number to divide by
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] >= 1000.0                                                        # north_star's floor


def test_roofline_object():
    r = latest_line()["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / average launch duration (HIP events minus the empty-pair time)
    gbs = r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9
    assert abs(gbs - r["achieved"]) < 0.01 * r["achieved"]
    assert r["event_pair_us"] > r["avg_launch_us"] > 0 and r["empty_event_pair_us"] > 0
    assert r["traffic"] is None or 0 < r["traffic"] < 4 * r["algorithmic_bytes_per_launch"]
    # 80 B per sampled model pixel and executed iteration, at 640x480 with the recorded iteration counts
    it = r["icp_iterations_per_level"]
    px = [307200, 76800, 19200]
    assert r["algorithmic_bytes_per_frame"] == 80 * sum(a * b for a, b in zip(it, px))


def test_cpu_baselines_and_probe():
    d = latest_line()
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "frames/s" and c["value"] > 0 and c["sample"]
    a = d["cpu_baseline_all_cores"]
    assert a["kind"] == "port" and a["cores"] > 1 and a["value"] > c["value"]
    p = d["hbm_stream_probe"]
    assert 1000.0 < p["copy"] < 8000.0 and 1000.0 < p["fill"] < 8000.0 and 1000.0 < p["read"] < 8000.0
    for k in d["hbm_bound_kernels"]:
        assert abs(k["frac"] - k["achieved"] / 8000.0) < 1e-3 and 0 < k["frac_of_measured_stream"] < 1.5
        gbs = k["algorithmic_bytes_per_launch"] / (k["avg_launch_us"] * 1e-6) / 1e9
        assert abs(gbs - k["achieved"]) < 0.02 * k["achieved"]
