"""Child process of tests/test_rccl_collate.py (never imported by pytest: no test_ prefix).

Forms a world-size-1 `nccl` (= RCCL) process group on cuda:0 and runs the collate step of bench.py's N > 1 path exactly as bench.py
issues it (shard.collate_after): rtd_infer_async -> rtd_signal_stream(torch-owned stream) -> all_gather_into_tensor over the zero-copy
view of rtd_result_block on that stream -> rtd_wait_stream.  Prints one JSON line; exit code 0 = the gathered block equals the result
block bit for bit on every repetition."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "MASTER_PORT" not in os.environ:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.shard import DevBlock, cameras_of_rank, collate_after
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    arch = ARCHS["r18"]
    B, H, Q = 2, 640, arch.num_queries
    eng = _capi.Engine(arch, pack_blob(fold_weights(arch, synth_weights(arch, 0))), device=0, precision=_capi.PREC_F16X3, max_batch=B,
                       input_size=(H, H), use_graph=True)
    stream = torch.cuda.Stream(device=torch.device("cuda", 0))      # torch's own: the engine's stream never reaches torch or RCCL
    gathered = torch.full((dist.get_world_size() * B * Q * 6,), float("nan"), dtype=torch.float32, device="cuda")
    ok, reps = True, 4
    for rep in range(reps):
        frames = [torch.from_numpy(scene_frame(900 + 2 * rep, H, H)).cuda(), torch.from_numpy(noise_frame(901 + 2 * rep, H, H)).cuda()]
        torch.cuda.synchronize()
        args = eng.make_async_args(frames)
        eng.infer_async_prepared(args)
        collate_after(eng, gathered, stream)                 # as in bench.py
        if rep % 2 == 1:
            eng.infer_async_prepared(args)                   # the next forward is ordered behind the all-gather: same frames, same block
        eng.sync()
        torch.cuda.synchronize()
        ptr, n = eng.result_block()
        block = torch.as_tensor(DevBlock(ptr, n), device="cuda:0")
        want = block.clone().cpu().numpy()
        got = gathered.cpu().numpy()
        # independent check of the block itself: the synchronous path on the same frames
        labels, boxes, scores = eng.infer_raw([f.cpu().numpy() for f in frames])
        ref = np.concatenate([labels[..., None].astype(np.float32), scores[..., None], boxes], -1).reshape(-1)
        ok = ok and np.array_equal(got, want) and np.array_equal(want, ref) and np.isfinite(got).all()
    out = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "bit_exact": bool(ok), "reps": reps, "floats": int(gathered.numel()),
           "cameras_of_rank0": cameras_of_rank(B, 0, 1)}
    print(json.dumps(out), flush=True)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
