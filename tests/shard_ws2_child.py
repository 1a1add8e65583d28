"""Child process of tests/test_rccl_collate.py::test_two_ranks_with_real_engines... (never imported by pytest: no test_ prefix).

One RANK of a world-size-2 `gloo` group (RANK / WORLD_SIZE / MASTER_* from the environment, as torch.distributed.run sets them), with a REAL
engine on cuda:0 - the one GPU of a test box is shared by the two processes.  BASELINE configs[3] in small: four cameras, camera k ->
rank k mod 2 (shard.cameras_of_rank), each rank runs its cameras as one batch, every rank contributes its [n, Q, 6] block to the
all-gather (host tensors: two ranks cannot form an RCCL communicator on ONE device), rank 0 turns the gathered blocks into the
reference's detection dicts (shard.block_to_detections) and compares them with what a single detector says about each camera's frame."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    import datetime
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    if os.environ.get("SHARD_DRY") == "1":                     # CPU rehearsal of the launch + rendezvous + gather (no engine)
        from telescope_cam_detection_amd.shard import cameras_of_rank, collate_blocks
        cams = cameras_of_rank(4, rank, world)
        g = collate_blocks(torch.stack([torch.full((5, 6), float(k)) for k in cams]))
        okd = all(float(g[r, j, 0, 0]) == k for r in range(world) for j, k in enumerate(cameras_of_rank(4, r, world)))
        if rank == 0:
            print(json.dumps({"world": world, "backend": dist.get_backend(), "dry": True, "bit_exact": bool(okd),
                              "cameras": {str(r): cameras_of_rank(4, r, world) for r in range(world)}}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return 0 if okd else 1
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.shard import DevBlock, block_to_detections, cameras_of_rank, collate_blocks
    from telescope_cam_detection_amd.synth import scene_frame
    from telescope_cam_detection_amd.weights import cached_blob, synth_weights

    arch = ARCHS["r18"]
    H, Q, n_cams = 320, arch.num_queries, 4
    blob = cached_blob(arch, "synthetic:r18:0", lambda: synth_weights(arch, 0))     # both ranks at once: the cache's atomic write
    cams = cameras_of_rank(n_cams, rank, world)
    frames = {k: scene_frame(500 + k, 300 + 20 * k, 400) for k in range(n_cams)}    # camera k's frame (its own size: the resampler runs)
    eng = _capi.Engine(arch, blob, device=0, precision=_capi.PREC_F16X3, max_batch=len(cams), input_size=(H, H), use_graph=True)
    ok = True
    for rep in range(2):
        eng.infer_async([frames[k] for k in cams], on_device=False)
        eng.sync()
        ptr, n = eng.result_block()
        block = torch.as_tensor(DevBlock(ptr, n), device="cuda:0").clone().view(len(cams), Q, 6).cpu()
        gathered = collate_blocks(block)                                            # [world, n, Q, 6] on every rank
        if rank == 0:
            for r in range(world):
                dets = block_to_detections(gathered[r].numpy(), 0.05, False)
                for j, k in enumerate(cameras_of_rank(n_cams, r, world)):
                    one = _capi.Engine(arch, blob, device=0, precision=_capi.PREC_F16X3, max_batch=1, input_size=(H, H), use_graph=False) if rep == 0 else None
                    if one is not None:
                        rows = one.infer([frames[k]], 0.05, False)[0]
                        one.close()
                        want = [{"class_id": int(x["class_id"]), "confidence": float(x["score"]),
                                 "bbox": {"x1": float(x["x1"]), "y1": float(x["y1"]), "x2": float(x["x2"]), "y2": float(x["y2"])}} for x in rows]
                        got = [{"class_id": d["class_id"], "confidence": d["confidence"], "bbox": {c: d["bbox"][c] for c in ("x1", "y1", "x2", "y2")}} for d in dets[j]]
                        ok = ok and len(got) > 0 and got == want
        dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "backend": dist.get_backend(), "cameras": {str(r): cameras_of_rank(n_cams, r, world) for r in range(world)},
                          "bit_exact": bool(ok)}), flush=True)
    eng.close()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
