#!/usr/bin/env python3
"""Probe: does running two half-batches on two handles (= two HIP streams / graphs) beat one bs-8 handle?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.synth import noise_frame
from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

arch = ARCHS["r50"]
blob = pack_blob(fold_weights(arch, synth_weights(arch, 0)))
frames = [torch.from_numpy(noise_frame(2000 + i, 640, 640)).cuda() for i in range(8)]

def run(split, steps=60, warm=8):
    engs = [_capi.Engine(arch, blob, 0, _capi.PREC_BF16, 8 // split, (640, 640), True) for _ in range(split)]
    args = [e.make_async_args(frames[i * (8 // split):(i + 1) * (8 // split)]) for i, e in enumerate(engs)]
    for _ in range(warm):
        for e, a in zip(engs, args): e.infer_async_prepared(a)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, a in zip(engs, args): e.infer_async_prepared(a)
    for e in engs: e.sync()
    dt = time.perf_counter() - t0
    for e in engs: e.close()
    return 8 * steps / dt, 1e3 * dt / steps

for split in (1, 2, 4, 1, 2):
    fps, ms = run(split)
    print(f"handles={split} (bs={8 // split} each): {fps:8.1f} frames/s  {ms:.3f} ms per 8 frames", flush=True)


def run_full(nh, steps=60, warm=8):
    """nh handles with a FULL bs-8 batch each, issued round-robin (nh batches in flight)"""
    engs = [_capi.Engine(arch, blob, 0, _capi.PREC_BF16, 8, (640, 640), True) for _ in range(nh)]
    args = [e.make_async_args(frames) for e in engs]
    for _ in range(warm):
        for e, a in zip(engs, args): e.infer_async_prepared(a)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, a in zip(engs, args): e.infer_async_prepared(a)
    for e in engs: e.sync()
    dt = time.perf_counter() - t0
    for e in engs: e.close()
    return 8 * nh * steps / dt, 1e3 * dt / (steps * nh)

for nh in (1, 2, 3, 1, 2):
    fps, ms = run_full(nh)
    print(f"full-batch handles={nh}: {fps:8.1f} frames/s  {ms:.3f} ms per 8-frame step (throughput)", flush=True)
