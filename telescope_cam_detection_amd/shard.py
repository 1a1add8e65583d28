"""Camera sharding across GPUs: one process per GPU, camera k -> rank k mod world (mirrors the reference's
one-engine-per-camera layout, /root/reference/main.py:1236-1279), and the ONE exchange step of the path:
an all-gather of every rank's fixed-size detection block so rank 0 (the web server process,
main.py:1223,1287-1291) sees all cameras.  Payload per rank: [n_cameras_local, Q, 6] fp32 rows
(label, score, x1, y1, x2, y2) = 7.2 KB per camera - latency-bound.  It runs on a stream torch owns, ordered behind the forward by an
event of the engine (`collate_after`): torch and RCCL never see the engine's own stream.

`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is plumbing only.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from .coco_constants import COCO_CLASSES, WILDLIFE_CLASSES


def camera_rank(camera_index: int, world_size: int) -> int:
    return camera_index % world_size


def cameras_of_rank(n_cameras: int, rank: int, world_size: int) -> List[int]:
    return [k for k in range(n_cameras) if camera_rank(k, world_size) == rank]


def collate_blocks(block, out=None, group=None):
    """all_gather_into_tensor of one [n, Q, 6] block per rank -> [world, n, Q, 6] on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out.view(-1), block.contiguous().view(-1), group=group)
    return out


class DevBlock:
    """zero-copy `__cuda_array_interface__` view of the engine's result block (rtd_result_block): [n_floats] fp32 in HBM"""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def collate_after(engine, out, comm_stream, group=None):
    """The collate step of one rank: all-gather the engine's result block of the batch just submitted.

    `comm_stream` is a torch-owned stream.  It waits for the engine's forward through an event of the LIBRARY (rtd_signal_stream), runs
    the all-gather, and the engine's stream then waits for it (rtd_wait_stream) so the next forward cannot overwrite the block while RCCL
    still reads it.  No torch / RCCL event is ever recorded on the engine's stream."""
    import torch

    ptr, n = engine.result_block()
    block = torch.as_tensor(DevBlock(ptr, n), device=out.device)
    engine.signal_stream(comm_stream.cuda_stream)
    with torch.cuda.stream(comm_stream):
        collate_blocks(block, out=out, group=group)
    engine.wait_stream(comm_stream.cuda_stream)
    return out


def block_to_detections(block: np.ndarray, conf_threshold: float = 0.25, wildlife_only: bool = True) -> List[List[Dict]]:
    """[n, Q, 6] rows -> per-camera detection dicts with the schema of src/rtdetr_detector.py:290-301
    (rows are already in descending score order)."""
    out = []
    for cam in np.asarray(block):
        dets = []
        for lab, score, x1, y1, x2, y2 in cam:
            score = float(score)
            if score < conf_threshold:
                continue
            cid = int(lab)
            if wildlife_only and cid not in WILDLIFE_CLASSES:
                continue
            x1, y1, x2, y2 = float(x1), float(y1), float(x2), float(y2)
            dets.append({"class_id": cid, "class_name": COCO_CLASSES[cid] if cid < len(COCO_CLASSES) else f"class_{cid}",
                         "confidence": score, "bbox": {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "area": int((x2 - x1) * (y2 - y1))}})
        out.append(dets)
    return out
