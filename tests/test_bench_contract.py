"""The bench line's contract.  The checks live in bench.check_line, which bench.py applies to its own output before it
prints; here the same function runs (CPU) on the line committed with the round's artefacts -- a format check of the
artefact -- and (GPU) on a live `python bench.py` run, so a regression in bench.py itself cannot hide behind an old file."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line"
    text = open(files[-1]).read().strip()
    assert "\n" not in text, "bench.py prints ONE line"
    return json.loads(text)


def test_committed_line_passes_the_contract():
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = latest_line()
    d["config"].setdefault("rows", 480)              # lines of round 1 did not carry the frame size
    d["config"].setdefault("cols", 640)
    bench.check_line(d, base["metric"])
    assert d["n_gpus"] == 1 and d["value"] >= 1000.0 and base["published"] == {}      # north_star's floor; nothing published to divide by
    c, a = d["cpu_baseline"], d["cpu_baseline_all_cores"]
    assert c["cores"] == 1 and a["cores"] > 1 and a["value"] > c["value"]
    p = d["hbm_stream_probe"]
    assert 1000.0 < p["copy"] < 8000.0 and 1000.0 < p["fill"] < 8000.0 and 1000.0 < p["read"] < 8000.0


def test_check_line_rejects_broken_lines():
    import bench
    d = latest_line()
    d["config"].setdefault("rows", 480)
    d["config"].setdefault("cols", 640)
    for breakage in (lambda x: x.pop("roofline"), lambda x: x.__setitem__("value", x["value"] * 2),
                     lambda x: x["roofline"].__setitem__("frac", 0.9), lambda x: x["config"].__setitem__("model", "m")):
        e = json.loads(json.dumps(d))
        breakage(e)
        with pytest.raises((AssertionError, KeyError)):
            bench.check_line(e)


@pytest.mark.gpu
def test_live_bench_run_passes_the_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--cpu-frames", "2",
                        "--no-large-kernels"], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    import bench
    d = bench.check_line(json.loads(lines[0]), json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"])
    assert d["roofline"]["kernel"] == "k_icp_loop" and d["roofline"]["traffic_source"]["file"].startswith("profiles/")
    assert d["value_with_h2d"]["value"] > 500 and d["cpu_baseline"]["cores"] == 1
    q = d["parity_in_run"]                           # 3 x 2 oracle frames: the 5 warm-up frames and the first timed one
    assert q["bit_exact"] is True and q["frames"] == 6 and q["timed_frames"] == 1 and q["max_abs_pose_diff"] == 0.0
