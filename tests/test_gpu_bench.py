"""bench.py end to end on the GPU box: the line the driver parses must come out for the default configuration (C3, with
the exact-fp32 parity leg) and for the chain configuration (C2), with the objects the contract names.

Schema and internal consistency only.  Throughput is what bench.py REPORTS, never what a correctness suite asserts: a
3-step run on a box that has just been leased measures the cold start, not the kernels (round 2's driver run: 317 858
samples/s here against 5.93 M in the driver's own bench), and these tests are collected LAST (conftest.py) so that
nothing end-to-end can stand in front of the parity tests under `pytest -x`."""
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
    d = json.loads(lines[0])
    print("bench line:", d["ms_per_step"], "ms/step; enqueue", d["host_enqueue_ms_per_step"], d.get("host_enqueue_done_ms_first_steps"),
          "ramp", d.get("ramp_up_step_ms"))
    return d, lines[0]


def test_default_bench_line_c3_with_fp32_parity_leg():
    d, line = _run()                               # exactly what the driver runs, fewer steps; cpu_baseline on a bounded sample
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "bf16" and d["higher_is_better"] is True, line
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["vs_baseline"] is None, line
    assert abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"], line
    assert d["warmup"] >= 1 and d["warmup_effective"] >= d["warmup"], line
    assert d["step_path"] == "layers" and d["config"]["global_batch"] == 8192, line
    assert "workload" in d["config"] and "model" not in d["config"], line
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9, line
    by = r["by_kernel"]
    assert {"gemm_fwd", "gemm_dgrad", "gemm_wgrad", "loss", "adam"} <= set(by), line
    # the kernel the roofline is quoted on is the class with the largest share of the step among the GEMM classes
    gemm = [k for k in by if k.startswith("gemm_")]
    assert r["kernel"] == max(gemm, key=lambda k: by[k]["ms_per_step"]), line
    assert by[r["kernel"]]["sampled_steps"] >= 3, line
    rs = d["roofline_step"]
    assert rs["bound"] == "mfma" and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-9 and 0.0 < rs["frac"] < 1.0, line
    f = d["f32_parity"]
    assert f["ms_per_step"] > 0 and f["steps"] == 10 and abs(f["samples_per_s"] - 8192 / (f["ms_per_step"] * 1e-3)) <= 1e-6 * f["samples_per_s"], line
    assert abs(f["frac"] - f["tflops"] / f["roofline_tflops"]) < 1e-9 and f["roofline_tflops"] in (157.3, 2500.0 / 6.0), line
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1, line
    # thread pools capped at the container's CPU quota before the timed region (codae/hostcpu.py): the baseline ran on that many
    assert d["host_cpu_share"] >= 1 and d["host_threads"] <= d["host_cpu_share"] and c["cores"] <= d["host_cpu_share"], line


def test_bench_line_c2_takes_the_chain():
    d, line = _run("--config", "c2", "--no-f32-parity", "--no-cpu-baseline")
    assert d["step_path"] == "chain" and d["config"]["global_batch"] == 1024, line
    assert "chain" in d["roofline"]["by_kernel"] and d["value"] > 0, line
    assert d["roofline"]["bound"] == "hbm", line
