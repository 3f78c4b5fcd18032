"""bench.py's last stdout line is what the driver records: it must stay a small, parseable JSON object whatever the detail
behind it grows to (round 5's line reached 22 KB and the driver's record lost it), and `--gpus N` must mean N ranks."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_driver*.json")))


def _detail(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_line_from_a_recorded_detail_is_small_and_round_trips(path):
    d = _detail(path)
    if "detail" in d:      # a recorded LINE (round 6 on: profiles/r06*_bench_driver.json; the detail object is the *_detail.json beside it)
        raw = open(path).read().strip().splitlines()[-1]
        assert len(raw) < bench.LINE_LIMIT and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["cores"] > 0
    line = bench.compact_line(d)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["value"] == pytest.approx(d["value"], rel=1e-5) and j["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-5)
    assert j["config"]["workload"] and j["config"]["baseline_config"] == d["config"]["baseline_config"]
    if d.get("roofline"):
        assert j["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-5)
        assert j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0
        assert "launch_us" not in j["roofline"]
    if d.get("cpu_baseline"):
        assert j["cpu_baseline"]["cores"] == d["cpu_baseline"]["cores"] and j["cpu_baseline"]["kind"] == "reference"
    for name, c in (j.get("configs") or {}).items():
        assert not any(isinstance(v, list) for v in c.values()), name       # flat: no per-window or per-domain arrays


def test_line_survives_a_detail_far_larger_than_any_recorded():
    d = _detail(RECORDED[-1])
    d["roofline"]["launch_us"] = [441.0] * 5000
    d["config"]["workload"] = "w" * 5000
    d["error"] = "e" * 5000
    d["configs"] = {f"c{i}": dict(d["configs"]["c5"]) for i in range(3, 40)}     # would not fit: dropped, the contract fields stay
    line = bench.compact_line(d)
    assert len(line) <= bench.LINE_LIMIT
    j = json.loads(line)
    assert j["value"] == pytest.approx(d["value"], rel=1e-5) and j["roofline"] and j["cpu_baseline"]


def test_plain_gpus_n_without_n_devices_fails_loudly():
    """`python bench.py --gpus 8` (no WORLD_SIZE) launches 8 ranks itself; with fewer devices it says so instead of
    quietly running one rank (here: no GPU at all)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MZR_BENCH_SINGLE_DEVICE")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 8" in r.stderr and "GPU(s) visible" in r.stderr


def test_gpus_must_agree_with_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr
