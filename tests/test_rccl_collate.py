"""GPU: the RCCL collate step (the reference's shared detection queue, main.py:1236-1291) executed for real on the one GPU
a test box has: a world-size-1 `nccl` process group in a fresh child process (the test session's own torch state stays untouched)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_all_gather_of_the_result_block_on_the_engine_stream_world_size_1():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_ws1_child.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    print(out)
    assert out["rccl_ranks"] == 1 and out["backend"] == "nccl" and out["bit_exact"] is True and out["floats"] == 2 * 300 * 6
    assert out["cameras_of_rank0"] == [0, 1]
