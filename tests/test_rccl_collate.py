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


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_two_ranks(extra_env, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "shard_ws2_child.py")], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_ranks_with_real_engines_shard_four_cameras_and_collate():
    """The N > 1 data path with REAL engines (BASELINE configs[3] in small; the 8-GPU curve itself is the driver's to measure): two ranks
    launched by torch.distributed.run share the box's one GPU, camera k -> rank k mod 2, one batch per rank, an all-gather of the fixed
    [n, Q, 6] blocks (gloo: two ranks cannot form an RCCL communicator on one device - the RCCL collective itself is the world-size-1 test
    above), rank 0 formats every camera's detections: equal, dict for dict, to a single engine's answer for that camera's frame."""
    out = run_two_ranks({}, 300)
    print(out)
    assert out["world"] == 2 and out["backend"] == "gloo" and out["bit_exact"] is True and out["cameras"] == {"0": [0, 2], "1": [1, 3]}
