"""bench.py prints ONE JSON line with the driver's contract keys (run at the small workload so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "dtu_640x480_v2_it4", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert d["config"]["workload"] == "dtu_640x480_v2_it4" and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
