"""bench.py as the driver runs it - `python bench.py --gpus 1 --steps K --warmup W`, every leg switched on - must finish and print ONE JSON
line with the contract's fields (a NameError in the CPU-baseline leg once survived a whole round because nothing ran that leg before the
driver did).  Also a second, lean line for the two_stage workload and the collate step (world-size-1 RCCL group)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*args, timeout=600):
    env = dict(os.environ, RTD_CPU_THREADS=os.environ.get("RTD_CPU_THREADS", "16"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_command_prints_the_contract_line_with_every_leg():
    d = run_bench("--gpus", "1", "--steps", "10", "--warmup", "3")
    assert d["metric"] == "frames_per_sec" and d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f16x3"
    assert d["value"] > 500 and abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3
    assert "RT-DETR-R50 640x640 bs=8" in d["config"]["workload"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.03 < rf["frac"] < 0.4
    # the committed PMC / rocprofv3 summaries must belong to THESE kernel sources (tools/refresh_profiles.sh + commit_profiles.sh after the
    # last kernel change of a round): bench.py attaches them only when their csrc stamp matches
    assert rf["traffic"] is not None and rf["traffic"] > 10 and "avg_launch_us_rocprofv3" in rf and "mfma_busy_frac_pmc" in rf, rf
    assert abs(rf["avg_launch_us_rocprofv3"] - rf["avg_launch_us"]) / rf["avg_launch_us"] < 0.15      # HIP-event and rocprofv3 durations agree
    assert d["hbm"]["gbytes_per_step_pmc"] > 5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "frames/s" and "sample" in cb
    assert d["cpu_baseline_r18_bs1"]["p50"] > 0 and d["cpu_baseline_r18_bs1"]["kind"] == "port"
    assert d["p50_ms_per_frame_bs1"] < 10.0                              # the north star's latency target
    assert d["rccl_ranks"] == 0 and "detect_host_ms" in d and "multi_stream" in d and "bf16_engine" in d


def test_two_stage_and_collate_lines():
    lean = ("--steps", "10", "--warmup", "3", "--multi-streams", "0", "--no-bf16-line", "--no-cpu-baseline", "--no-latency", "--no-detect-host", "--no-mfma-probe")
    d = run_bench("--workload", "two_stage", *lean)
    assert d["config"]["stage2"]["crops_per_step"] == 16 and "asynchronous" in d["config"]["stage2"]["timed"] and d["value"] > 500
    d = run_bench("--collate", "--batch", "1", *lean)                    # BASELINE configs[3]'s per-rank shape at N = 1
    assert d["rccl_ranks"] == 1 and d["collate"]["bytes_per_rank"] == 300 * 6 * 4 and d["collate"]["rccl_ranks"] == 1
    assert "bs=1/GPU" in d["config"]["workload"]
