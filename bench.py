#!/usr/bin/env python3
"""bench.py - frames/s of the MI355X-native RT-DETR path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): RT-DETR-R50, 640x640, bs=8 per GPU, bf16, synthetic uniform-noise
uint8 BGR frames (the reference benchmark's input, tests/test_inference.py:76) already resident in HBM,
seeded synthetic weights (no checkpoint exists offline).  One "step" = one batch of 8 frames through
preprocess -> network -> post-process on the device (the fixed [8,300,6] result block stays in HBM).
N > 1: one process per GPU, cameras sharded over ranks (weak scaling), and one RCCL all-gather of
every rank's result block per step - the collate step for rank 0's web server.

--streams S (default 3): S engine handles per GPU, each with its own HIP stream, hipGraph and camera group, take the K
timed steps round-robin, so S batches are in flight - the reference's deployment shape (one inference engine per camera
group sharing the GPU, main.py:1236-1291).  Every step is still one full bs-8 pass; the kernels of one batch fill the CUs
the other batches' small grids and launch ramps leave idle (same-box: S = 1 / 2 / 3 / 4 -> 2256 / 2930 / 3127 / 2868 frames/s).
`single_stream` in the JSON is the same measurement with S = 1.

Prints ONE JSON line on rank 0 with the contract fields plus:
  roofline     - MFMA roofline of the dominant kernel family (conv_igemm), from HIP-event timings of
                 every launch on the engine's stream (rtd_profile) and the algorithmic FLOPs per launch
  cpu_baseline - the CPU oracle (oracle/rtdetr_oracle.py, fp32 eager PyTorch) timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0, "fp32": 157.3}   # /opt/skills/guides/MI355X_MICROARCH.md (dense)
HBM_PEAK_GBS = 8000.0
CANON_GFLOP_PER_FRAME = {"r18": 60.53, "r50": 133.91}  # BASELINE.md §3 @640x640


class _DevPtr:
    """zero-copy torch view of a raw device pointer (the engine's result block)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="r50")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--precision", default="bf16", choices=["bf16x3", "bf16", "fp32"])
    ap.add_argument("--streams", type=int, default=3, help="engine handles (batches in flight) per GPU")
    ap.add_argument("--latency-profile", action="store_true", help="A/B: the handles of a multi-handle run use the latency profile too")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--profile-out", default="")
    ap.add_argument("--opt", action="append", default=[], help="rtd_debug_option name=value (A/B runs)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the hot path"
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.shard import collate_blocks
    from telescope_cam_detection_amd.synth import noise_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    for o in args.opt:
        name, val = o.split("=")
        _capi.debug_option(name, int(val))
    arch = ARCHS[args.arch]
    B, H = args.batch, args.size
    w = synth_weights(arch, 0)
    blob = pack_blob(fold_weights(arch, w))
    prec = _capi.precision_code(args.precision)
    S = max(1, args.streams)
    # several handles per GPU run the throughput profile (rtd_config.profile: 256-pixel conv tiles); the one-batch-in-flight
    # figure, the kernel profile / roofline and the latency line come from a latency-profile handle, which is also what
    # `--streams 1` (the rocprofv3 runs under profiles/) measures
    prof_mode = _capi.PROFILE_THROUGHPUT if (S > 1 and not args.latency_profile) else _capi.PROFILE_LATENCY
    engs = [_capi.Engine(arch, blob, device=local_rank, precision=prec, max_batch=B, input_size=(H, H), use_graph=not args.no_graph,
                         profile=prof_mode) for _ in range(S)]
    eng = engs[0]
    eng_lat = eng
    if prof_mode != _capi.PROFILE_LATENCY:
        eng_lat = _capi.Engine(arch, blob, device=local_rank, precision=prec, max_batch=B, input_size=(H, H), use_graph=not args.no_graph)
    # frame i of config c = default_rng(1000*c+i).integers(0,255,(H,W,3),uint8); camera k -> rank k
    frames_of = [[torch.from_numpy(noise_frame(2000 + (rank * S + si) * B + i, H, H)).cuda() for i in range(B)] for si in range(S)]
    frames = frames_of[0]
    prepared = [e.make_async_args(f) for e, f in zip(engs, frames_of)]
    Q = arch.num_queries

    streams = [torch.cuda.ExternalStream(e.stream(), device=torch.device("cuda", local_rank)) for e in engs]
    gathered = None
    if world > 1:
        gathered = [torch.empty(world * B * Q * 6, dtype=torch.float32, device="cuda") for _ in range(S)]

    prepared_lat = eng_lat.make_async_args(frames) if eng_lat is not eng else prepared[0]

    def step(k, n_handles):
        si = k % n_handles
        if n_handles == 1 and world == 1:        # one batch in flight: the latency-profile handle
            eng_lat.infer_async_prepared(prepared_lat)
            return
        engs[si].infer_async_prepared(prepared[si])
        if world > 1:
            ptr, n = engs[si].result_block()
            block = torch.as_tensor(_DevPtr(ptr, n), device=f"cuda:{local_rank}")
            with torch.cuda.stream(streams[si]):              # ordered after the forward on that engine's stream
                collate_blocks(block, out=gathered[si])

    def fence():
        for e in engs:
            e.sync()
        eng_lat.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_handles):
        for k in range(args.warmup):
            step(k, n_handles)
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k, n_handles)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    single = timed(1) if S > 1 else None
    elapsed = timed(S)
    fps = world * B * args.steps / elapsed
    out = {
        "metric": "frames_per_sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"RT-DETR-{args.arch.upper()} {H}x{H} bs={B}/GPU, uint8 BGR frames resident in HBM -> "
                               f"[{B},{Q},6] detections in HBM; synthetic seeded weights",
                   "global_batch": world * B, "parallelism": f"camera-shard x{world}" + (" + RCCL all_gather of detections" if world > 1 else ""),
                   "streams_per_gpu": S, "batches_in_flight_per_gpu": S, "kernel_profile": "throughput" if prof_mode == _capi.PROFILE_THROUGHPUT else "latency", "hip_graph": not args.no_graph},
    }
    if single is not None:
        out["single_stream"] = {"value": round(world * B * args.steps / single, 2), "unit": "frames/s",
                                "ms_per_step": round(1000.0 * single / args.steps, 4),
                                "note": "same K steps on ONE latency-profile handle (one batch in flight); ms_per_step here is the latency of one bs-" + str(B) + " step"}
    if args.arch in CANON_GFLOP_PER_FRAME and H == 640:
        out["mfma_frac_whole_model"] = round(CANON_GFLOP_PER_FRAME[args.arch] * fps / world / (MFMA_PEAK_TFLOPS[args.precision] * 1e3), 4)

    if rank == 0:
        # ---- per-kernel HIP-event profile on the engine's stream -> roofline of the dominant kernel ----
        print(f"[bench] {fps:.1f} frames/s; profiling kernels ...", file=sys.stderr, flush=True)
        prof = eng_lat.profile(B, reps=5)
        fam = {}
        for p in prof:
            f = fam.setdefault(p["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            f["ms"] += p["ms"]; f["flops"] += p["flops"]; f["bytes"] += p["bytes"]; f["launches"] += 1
        total_ms = sum(f["ms"] for f in fam.values())
        dom = max(fam, key=lambda k: fam[k]["ms"])
        d = fam[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.precision]
        out["roofline"] = {
            "kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
            "alg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
            "alg_gbytes_per_s_unfused": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
            "share_of_step": round(d["ms"] / total_ms, 3),
            "method": "rtd_profile: hipEvent pairs around every launch of one eager forward on ONE engine stream (no second batch in flight), mean of 5",
        }
        # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
        # same command; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md) - counters cannot be read live
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as fh:
                pmc = json.load(fh)
            if dom == "conv_igemm" and args.arch == "r50" and B == 8 and H == 640 and args.precision == "bf16":
                fams = [v for k, v in pmc.items() if isinstance(v, dict) and k.startswith("conv")]   # LDS-DMA, direct 3x3 and v1 kernels
                tb = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in fams)
                nl = sum(v["launches"] for v in fams)
                out["roofline"]["traffic"] = round(tb / nl / 1e6, 2)
                out["roofline"]["traffic_unit"] = "MB HBM per launch (PMC, profiles/r01_pmc_hbm_traffic.json)"
                out["roofline"]["alg_mbytes_per_launch_unfused"] = round(d["bytes"] / d["launches"] / 1e6, 2)
                # cross-check: the committed rocprofv3 --kernel-trace of `--streams 1` (hipGraph replay).  Its durations are
                # kernel execution only; the HIP-event pairs above also contain the ~3-4 us between two dependent launches.
                with open(os.path.join(ROOT, "profiles", "r01_rocprofv3_kernel_summary.json")) as fh:
                    rp = json.load(fh)
                out["roofline"]["avg_launch_us_rocprofv3"] = rp["conv_igemm_all"]["avg_us"]
                out["roofline"]["achieved_rocprofv3"] = round(d["flops"] / (rp["conv_igemm_all"]["us_per_step"] * 1e-6) / 1e12, 2)
                # MFMA-pipe occupancy as the counters report it (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE x SIMDs): independent of the
                # clock the chip holds under load, low-biased on short dispatches (tools/summarize_profiles.py mfma)
                with open(os.path.join(ROOT, "profiles", "r01_pmc_mfma_util.json")) as fh:
                    mu = json.load(fh)
                out["roofline"]["mfma_busy_frac_pmc"] = mu["conv_igemm_all"]["mfma_util"]
        except (OSError, KeyError, ValueError):
            pass
        out["kernel_families_ms"] = {k: round(v["ms"], 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        if args.profile_out:
            with open(args.profile_out, "w") as fh:
                json.dump(prof, fh, indent=0)
        # ---- p50 single-frame latency (BASELINE metric's second half), synchronous detect() semantics ----
        if not args.no_latency and world == 1:
            print("[bench] bs=1 latency ...", file=sys.stderr, flush=True)
            lat = []
            one = [frames[0]]
            for i in range(60):
                t1 = time.perf_counter()
                eng_lat.infer(one, 0.25, True, on_device=True)
                if i >= 10:
                    lat.append((time.perf_counter() - t1) * 1e3)
            out["p50_ms_per_frame_bs1"] = round(float(np.percentile(lat, 50)), 4)
            out["p99_ms_per_frame_bs1"] = round(float(np.percentile(lat, 99)), 4)
        # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
        if not args.no_cpu_baseline and world == 1:
            from oracle import rtdetr_oracle as orc
            # the box's CPU share, not the host's core count (oversubscribed OpenMP spin-waits look like a hang)
            cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("RTD_CPU_THREADS", "16")))
            torch.set_num_threads(cores)
            print(f"[bench] cpu baseline on {cores} threads ...", file=sys.stderr, flush=True)
            nb = 2
            host = [f.cpu().numpy() for f in frames[:nb]]
            orc.detect_batch(arch, w, host, (H, H))          # warm-up
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                orc.detect_batch(arch, w, host, (H, H))
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(nb * reps / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"CPU oracle (fp32 eager PyTorch restatement of the reference path), RT-DETR-{args.arch.upper()} "
                                             f"{H}x{H} bs={nb}, {reps} timed batches after 1 warm-up, torch threads={cores}"}
        print(json.dumps(out), flush=True)
    for e in engs:
        e.close()
    if eng_lat is not eng:
        eng_lat.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
