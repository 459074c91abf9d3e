"""bench.py end to end on the GPU box: the line the driver parses must come out for the default configuration (C3, with
the exact-fp32 parity leg) and for the chain configuration (C2), with the objects the contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *flags],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_default_bench_line_c3_with_fp32_parity_leg():
    d = _run()                                     # exactly what the driver runs, fewer steps; cpu_baseline on a bounded sample
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "bf16" and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["value"] > 1e6 and d["vs_baseline"] is None
    assert d["step_path"] == "layers" and d["config"]["global_batch"] == 8192
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.2 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert {"gemm_fwd", "gemm_dgrad", "gemm_wgrad", "loss", "adam"} <= set(r["by_kernel"])
    assert d["f32_parity"]["ms_per_step"] > d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1


def test_bench_line_c2_takes_the_chain():
    d = _run("--config", "c2", "--no-f32-parity", "--no-cpu-baseline")
    assert d["step_path"] == "chain" and d["config"]["global_batch"] == 1024
    assert "chain" in d["roofline"]["by_kernel"] and d["value"] > 1e6
