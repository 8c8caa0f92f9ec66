"""bench.py's output contract on a small workload (GPU): one JSON line with the keys the driver
reads, `roofline` and `cpu_baseline` objects, for both tools."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(extra, base=("--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "12", "--sustain-seconds", "0.05")):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + list(base) + extra
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    text = out.stdout.decode()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    # the contract object is the LAST line of stdout and short enough for any log tail (the 20 KB
    # line of round 5 did not parse in the driver)
    assert text.rstrip("\n").splitlines()[-1] == lines[0]      # (also behind RCCL's banner: bench_side.emit flushes C stdio first)
    assert len(lines[0]) < 4096, len(lines[0])
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_of_the_drivers_exact_command():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- every side leg included, nothing skipped: the last stdout
    line is the contract object alone (< 4 KB, json.loads), carries roofline.frac and cpu_baseline.value, one number
    per side leg, and names the file the rest went to."""
    d = _run([], base=("--gpus", "1", "--steps", "20", "--warmup", "5"))
    for k in REQUIRED:
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel_ms"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1
    assert "extras_error" not in d, d.get("extras_error")
    side = d["side"]
    for k in ("variant422", "raw28", "device_stream", "sizes", "field_call", "field_submit", "field_submit422"):
        assert k in side, k
    full = json.load(open(os.path.join(ROOT, d["extras_file"])))
    assert full["value"] == pytest.approx(d["value"], rel=1e-4) and "end_to_end" in full
    for v in d.values():          # no prose: every string of the line is short
        assert not isinstance(v, str) or len(v) < 200


@pytest.mark.gpu
def test_bench_line_primary_tool():
    d = _run(["--cpu-fields", "6", "--cpu-mt-fields", "0", "--no-extras"])
    for k in REQUIRED:
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 24 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-4     # (the line carries 6 significant digits)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]
    assert r["algorithmic_bytes_per_launch"] == 8 * 720 * 243 * 24
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] in ("reference", "port") and cb["value"] > 0
    assert "workload" in d["config"] and d["value_sustained"] > 0


@pytest.mark.gpu
def test_bench_line_to_composite_tool():
    d = _run(["--tool", "to_composite", "--cpu-fields", "6"])
    for k in REQUIRED:
        assert k in d, k
    assert d["config"]["tool"] == "to_composite" and d["roofline"]["kernel"] == "k422_fused"
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 4 * 720 * 243 * 24
    assert d["cpu_baseline"]["value"] > 0 and d["value"] > 0


@pytest.mark.gpu
def test_bench_line_through_rccl_with_one_rank():
    """--force-dist: the barrier, the MAX all-reduce of the elapsed time and the all-gather of the rank
    checksums go through torch.distributed's nccl backend (= RCCL) even with a single rank, and rank 0's
    re-computation of every rank's checksum agrees.  (More ranks need more GPUs: tools/run_scaling.sh.)"""
    env_port = str(29600 + os.getpid() % 300)
    os.environ["MASTER_PORT"] = env_port
    try:
        d = _run(["--cpu-fields", "0", "--no-extras", "--force-dist", "--dist-backend", "nccl"])
    finally:
        os.environ.pop("MASTER_PORT", None)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["rank_checksums_verified"] is True and len(d["config"]["rank_checksums"]) == 1


@pytest.mark.gpu
def test_bench_config4_eight_streams_at_full_size_on_one_gpu():
    """BASELINE configs[3] at its real size -- 8 independent 300-frame 720x486 streams -- dealt stream s
    -> rank s % N; with the one GPU of this box N = 1, so rank 0 owns all eight (4,800 fields per step)."""
    d = _run(["--cpu-fields", "0", "--no-extras", "--streams", "8", "--frames", "300", "--steps", "2", "--warmup", "1"])
    assert d["config"]["fields_per_step_per_gpu"] == 8 * 600 and d["value"] > 0
    assert "8 independent 300-frame streams" in d["config"]["workload"]
