"""bench.py's output contract on a small scene (a child process, as the driver runs it): ONE JSON line on stdout with
the metric, the roofline object of the dominant kernel and the CPU baseline — from both meshlet sources."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("source", ["stream", "buffer"])
def test_bench_line_has_the_contract_fields(source):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--entities", "6000", "--cpu-sample-entities", "6000", "--cpu-seconds", "0.5",
                        "--settle-ms", "5", "--meshlet-source", source],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # libraries' banners go to stderr
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "meshlets/s" and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["meshlets_into_meshlet_cull"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    per = 20.25 if source == "stream" else 32  # the stream source with its alpha classes (bench.py default)
    assert rf["algorithmic_bytes_per_meshlet"] == per and rf["launches_timed"] >= 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["traffic"] is None  # the committed PMC passes are of the full-size scene only
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "meshlets/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["gpu_output_bit_identical"] is True and "sample" in cb
    assert ("derived meshlet streams" in d["config"]["meshlet_source"]) == (source == "stream")
