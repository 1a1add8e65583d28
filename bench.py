#!/usr/bin/env python3
"""bench.py - frames/s of the MI355X-native RT-DETR path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process touches no GPU and starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a child;
rank 0 of that world prints the JSON line.  Launched by torch.distributed.run directly (RANK / LOCAL_RANK / WORLD_SIZE set)
it runs as one rank.

Workload (BASELINE.json configs[1]): RT-DETR-R50, 640x640, bs=8 per GPU, synthetic uniform-noise uint8 BGR frames (the
reference benchmark's input, tests/test_inference.py:76) already resident in HBM, seeded synthetic weights (no checkpoint
exists offline).  One "step" = one batch of 8 frames through preprocess -> network -> post-process on the device (the
fixed [8,300,6] result block stays in HBM).  N > 1: one process per GPU, cameras sharded over ranks (weak scaling), and one
RCCL all-gather of every rank's result block per step - the collate step for rank 0's web server.

Engine (--precision): "f16x3" (default) = hi/lo fp16 pairs, three MFMAs per product, fp32 accumulate - the engine that meets
the north-star tolerance (1e-3 on scores, 1e-2 px on boxes; 640- and 1280-px frames, tests/test_gpu_parity.py) AND the throughput
target; "bf16" = plain bf16 storage / MFMA (faster, far outside the tolerance - printed as `bf16_engine` beside the headline);
"fp32" = exact fp32 MFMAs.

`value` is measured with ONE batch in flight (--streams 1: BASELINE's "bs=8" read strictly; ms_per_step is then the latency of a
step).  `multi_stream` repeats the measurement with --multi-streams S (default 3) engine handles per GPU, each with its own HIP
stream, hipGraph and camera group, taking the K timed steps round-robin - the reference's deployment shape (one inference engine
per camera group sharing the GPU, main.py:1236-1291); every step is still one full bs-8 pass.

--collate at N = 1: the RCCL collate step runs on a world-size-1 `nccl` group exactly as it does at N > 1 (same all_gather_into_tensor
over the zero-copy view of the result block, ordered behind the forward by the engine's event), `value` includes it and `collate` reports its cost per step.

Prints ONE JSON line on rank 0 with the contract fields plus:
  roofline     - MFMA roofline of the dominant kernel family (conv_igemm), from HIP-event timings of every launch on the
                 engine's stream (rtd_profile) and the algorithmic FLOPs per launch
  cpu_baseline - the CPU oracle (oracle/rtdetr_oracle.py, fp32 eager PyTorch) timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N = 1 only)
  detect_host_ms - the call the product makes (src/inference_engine_yolox.py:554): synchronous RTDETRDetector.detect(np.ndarray) on
                 HOST frames of 640x640, 1280x720 and 1920x1080 (input_size 640), 10 warm-ups + 100 timed calls each, R50 and R18:
                 mean / std / min / max / p50 / p95 / p99 - the protocol of the reference's tests/test_inference.py:63-115.  Includes
                 the H2D copy, the PIL-exact resampler, the D2H of the result block and the Python formatting; never `value`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16x3": 2500.0, "fp32": 157.3}   # /opt/skills/guides/MI355X_MICROARCH.md (dense)
MFMA_PER_PRODUCT = {"bf16": 1, "f16x3": 3, "fp32": 1}                  # MFMA flops issued per algorithmic flop
HBM_PEAK_GBS = 8000.0
CANON_GFLOP_PER_FRAME = {"r18": 60.53, "r50": 133.91}  # BASELINE.md 3 @640x640


def csrc_sha() -> str:
    """hash of the kernel sources: profiles/*.json carry the value they were measured at (tools/summarize_profiles.py)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "telescope_cam_detection_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="r50")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "fp32"])
    ap.add_argument("--streams", type=int, default=1, help="engine handles (batches in flight) per GPU behind `value`")
    ap.add_argument("--multi-streams", type=int, default=3, help="handles of the extra `multi_stream` measurement (0 = skip)")
    ap.add_argument("--workload", default="detect", choices=["detect", "two_stage"],
                    help="two_stage: detect + Stage-2 crop batch (BASELINE config 5; the classifier network itself is out of scope)")
    ap.add_argument("--latency-profile", action="store_true", help="A/B: the handles of a multi-handle run use the latency profile too")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-bf16-line", action="store_true", help="skip the secondary plain-bf16 engine measurement")
    ap.add_argument("--no-detect-host", action="store_true", help="skip the detect_host_ms measurement (RTDETRDetector.detect on host frames)")
    ap.add_argument("--no-mfma-probe", action="store_true", help="skip the sustained-MFMA-rate probe (roofline.sustained_mfma)")
    ap.add_argument("--collate", action="store_true", help="N = 1: run the RCCL collate step on a world-size-1 nccl group (always on at N > 1)")
    ap.add_argument("--profile-out", default="")
    ap.add_argument("--opt", action="append", default=[], help="rtd_debug_option name=value (A/B runs)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="CPU-only: form the world (gloo), all-gather the rank ids, print them as JSON - tests the self-launch path")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """--gpus N > 1 without a torch.distributed environment: start the N ranks as a child job.  Nothing in this process has
    touched the GPU (no torch.cuda call, no HIP library loaded); the child job's rank 0 prints the JSON line on our stdout."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def launch_selftest(args, rank, world) -> int:
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank], dtype=torch.int64)
    out = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(out, t)
    if rank == 0:
        print(json.dumps({"selftest": "launch", "world": dist.get_world_size(), "ranks": [int(x.item()) for x in out], "gpus": args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0



def bench_crop_step(batch: int, size: int, seed: int = 5000, n_crops: int = 16):
    """bench.py --workload two_stage (BASELINE config 5): after each detect step, the Stage-2 crop batch of `n_crops` mixed-size boxes
    (sides drawn from rng.integers(64, 512)) spread over the step's frames -> [16,3,336,336] classifier input, enqueued
    (asynchronously, rtd_crop_resize_batch) on a torch-owned stream that the engine's own event orders BEHIND this step's forward
    (rtd_signal_stream) - in the real pipeline the crops come from Stage-1 boxes; the next forward does not wait for
    the crops (it overwrites nothing they read), so Stage 2 of step k overlaps Stage 1 of step k + 1 as it would in a pipeline.  The classifier forward is excluded (its network is out of scope) and the boxes are synthetic: Stage 1 on noise
    frames with random weights finds nothing above threshold.  Returns (callable, description)."""
    import numpy as np
    import torch

    rng = np.random.default_rng(seed)
    from telescope_cam_detection_amd.stage2 import CropBatcher
    batcher = CropBatcher()
    rects = [[] for _ in range(batch)]
    for i in range(n_crops):
        cw, ch = int(rng.integers(64, min(512, size))), int(rng.integers(64, min(512, size)))
        x, y = int(rng.integers(0, size - cw + 1)), int(rng.integers(0, size - ch + 1))
        rects[i % batch].append((x, y, x + cw, y + ch))

    def run(engine, frames, stream):
        engine.signal_stream(stream.cuda_stream)          # Stage 2 after Stage 1
        with torch.cuda.stream(stream):
            batcher.preprocess_batch(frames, rects)

    info = {"crops_per_step": n_crops, "crop_sides": "rng.integers(64, 512)", "classifier_input": [n_crops, 3, batcher.input_size, batcher.input_size],
            "timed": "detect + crop/resize/normalise batch (one asynchronous launch on a torch-owned stream ordered after the forward); classifier forward EXCLUDED (EVA02 out of scope)"}
    return run, info

def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_selftest:
        sys.exit(launch_selftest(args, rank, world))

    import numpy as np
    import torch

    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the hot path"
    torch.cuda.set_device(local_rank)
    rccl_ranks = 0                                   # ranks of the nccl (= RCCL) group the collate step ran on; 0 = no group was formed
    collate = world > 1 or args.collate
    if collate:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        rccl_ranks = dist.get_world_size()

    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.shard import collate_after
    from telescope_cam_detection_amd.synth import noise_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

    for o in args.opt:
        name, val = o.split("=")
        _capi.debug_option(name, int(val))
    arch = ARCHS[args.arch]
    B, H = args.batch, args.size
    Q = arch.num_queries
    # the packed blob through the on-disk cache: the N ranks of a node fold the 43 M parameters once, not once each (weights.cached_blob)
    from telescope_cam_detection_amd.weights import cached_blob
    blob = cached_blob(arch, f"synthetic:{arch.name}:0", lambda: synth_weights(arch, 0))

    def fence(engs):
        for e in engs:
            e.sync()
        if rccl_ranks:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision: str, S: int, crops=None, collate=collate):
        """K timed steps round-robin over S engine handles of `precision`; returns (elapsed seconds, handles, their frames)."""
        prec = _capi.precision_code(precision)
        # several handles per GPU run the throughput profile (rtd_config.profile); a lone handle the latency profile
        prof_mode = _capi.PROFILE_THROUGHPUT if (S > 1 and not args.latency_profile) else _capi.PROFILE_LATENCY
        engs = [_capi.Engine(arch, blob, device=local_rank, precision=prec, max_batch=B, input_size=(H, H), use_graph=not args.no_graph,
                             profile=prof_mode) for _ in range(S)]
        # frame i of config c = default_rng(1000*c+i).integers(0,255,(H,W,3),uint8); camera k -> rank k
        frames_of = [[torch.from_numpy(noise_frame(2000 + (rank * S + si) * B + i, H, H)).cuda() for i in range(B)] for si in range(S)]
        prepared = [e.make_async_args(f) for e, f in zip(engs, frames_of)]
        # torch-owned streams for what torch / RCCL enqueue beside an engine (crop batch, all-gather); the engines' own streams stay
        # inside the library and are ordered against these through the library's events (shard.collate_after)
        streams = [torch.cuda.Stream(device=torch.device("cuda", local_rank)) for e in engs]
        gathered = [torch.empty(world * B * Q * 6, dtype=torch.float32, device="cuda") for _ in range(S)] if collate else None

        def step(k):
            si = k % S
            engs[si].infer_async_prepared(prepared[si])
            if crops is not None:
                crops(engs[si], frames_of[si], streams[si])
            if collate:
                collate_after(engs[si], gathered[si], streams[si])   # all-gather ordered after this engine's forward, next forward after it

        for k in range(args.warmup):
            step(k)
        fence(engs)
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        fence(engs)
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, engs, frames_of

    crops_fn, crop_info = None, None
    if args.workload == "two_stage":
        crops_fn, crop_info = bench_crop_step(B, H, seed=5000 + rank)

    S = max(1, args.streams)
    elapsed, engs, frames_of = measure(args.precision, S, crops_fn)
    eng, frames = engs[0], frames_of[0]
    fps = world * B * args.steps / elapsed
    out = {
        "metric": "frames_per_sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"RT-DETR-{args.arch.upper()} {H}x{H} bs={B}/GPU, uint8 BGR frames resident in HBM -> "
                               f"[{B},{Q},6] detections in HBM; synthetic seeded weights" + ("; + Stage-2 crop batch" if crop_info else ""),
                   "global_batch": world * B, "parallelism": f"camera-shard x{world}" + (" + RCCL all_gather of detections" if world > 1 else ""),
                   "streams_per_gpu": S, "batches_in_flight_per_gpu": S,
                   "kernel_profile": "throughput" if (S > 1 and not args.latency_profile) else "latency", "hip_graph": not args.no_graph,
                   "engine": {"f16x3": "hi/lo fp16 pairs, 3 MFMAs per product, fp32 accumulate (meets 1e-3 / 1e-2 px at 640 and 1280 px)",
                              "bf16": "bf16 storage + MFMA, fp32 accumulate (outside the 1e-3 / 1e-2 px tolerance)",
                              "fp32": "exact fp32 MFMAs"}[args.precision]},
        "rccl_ranks": rccl_ranks,
    }
    if collate:
        # the same K steps without the collate step: the all-gather's cost per step at THIS world size (world size 1: RCCL's launch + local copy)
        for e in engs:
            e.close()
        el0, e0, _ = measure(args.precision, S, crops_fn, collate=False)
        for e in e0:
            e.close()
        out["collate"] = {"ms_per_step_with": round(1000.0 * elapsed / args.steps, 4), "ms_per_step_without": round(1000.0 * el0 / args.steps, 4),
                          "all_gather_us_per_step": round(1e6 * (elapsed - el0) / args.steps, 1), "bytes_per_rank": B * Q * 6 * 4, "rccl_ranks": rccl_ranks,
                          "note": "torch.distributed all_gather_into_tensor (backend nccl = RCCL) over a zero-copy view of rtd_result_block, on a torch-owned stream ordered after the forward by the engine's own event (rtd_signal_stream / rtd_wait_stream)"}
        elapsed, engs, frames_of = measure(args.precision, S, crops_fn)       # handles for the profile / latency legs below
        eng, frames = engs[0], frames_of[0]
    if crop_info:
        out["config"]["stage2"] = crop_info
    if args.arch in CANON_GFLOP_PER_FRAME and H == 640:
        out["mfma_frac_whole_model"] = round(CANON_GFLOP_PER_FRAME[args.arch] * fps / world / (MFMA_PEAK_TFLOPS[args.precision] * 1e3), 4)

    if rank == 0:
        # ---- per-kernel HIP-event profile on the engine's stream -> roofline of the dominant kernel ----
        print(f"[bench] {fps:.1f} frames/s; profiling kernels ...", file=sys.stderr, flush=True)
        prof = eng.profile(B, reps=5)
        fam = {}
        for p in prof:
            f = fam.setdefault(p["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            f["ms"] += p["ms"]; f["flops"] += p["flops"]; f["bytes"] += p["bytes"]; f["launches"] += 1
        total_ms = sum(f["ms"] for f in fam.values())
        dom = max(fam, key=lambda k: fam[k]["ms"])
        d = fam[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.precision]
        mpp = MFMA_PER_PRODUCT[args.precision]
        out["roofline"] = {
            "kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "mfma_flops_per_alg_flop": mpp, "mfma_issue_frac": round(mpp * achieved / peak, 4),
            "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
            "alg_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
            "alg_gbytes_per_s_unfused": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
            "share_of_step": round(d["ms"] / total_ms, 3),
            "method": "rtd_profile: hipEvent pairs around every launch of one eager forward on ONE engine stream (no second batch in flight), mean of 5; "
                      "`achieved` counts ALGORITHMIC flops (2 per MAC); the f16x3 engine issues 3 MFMA flops per algorithmic flop (mfma_issue_frac)",
        }
        # HBM bytes per launch / rocprofv3 durations / MFMA-busy counters come from the committed PMC passes (separate rocprofv3 runs of
        # this same command: counters cannot be read live).  They are attached only when the profile was measured on THESE kernel sources
        # and THIS configuration (tools/refresh_profiles.sh writes them, tools/summarize_profiles.py stamps csrc hash + config).
        try:
            sha = csrc_sha()
            tag = f"{args.arch}_{H}_bs{B}_{args.precision}"
            sfx = "" if tag == "r50_640_bs8_f16x3" else f"_{args.arch}_{H}_bs{B}"
            import glob

            def newest(stem):                                       # profiles/rNN_<stem><sfx>.json of the latest round that has one
                found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}{sfx}.json")))
                if not found:
                    raise OSError(f"no profiles/rNN_{stem}{sfx}.json")
                return found[-1]
            pmc_path = newest("pmc_hbm_traffic")
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            if pmc.get("csrc_sha") == sha and pmc.get("config") == tag and dom == "conv_igemm":
                allf = [v for k, v in pmc.items() if isinstance(v, dict) and "launches" in v]
                fams = [v for k, v in pmc.items() if isinstance(v, dict) and k.startswith("conv")]
                tb = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in fams)
                nl = sum(v["launches"] for v in fams)
                out["roofline"]["traffic"] = round(tb / nl / 1e6, 2)
                out["roofline"]["traffic_unit"] = f"MB HBM per launch (PMC, profiles/{os.path.basename(pmc_path)} @ csrc {sha})"
                out["roofline"]["alg_mbytes_per_launch_unfused"] = round(d["bytes"] / d["launches"] / 1e6, 2)
                step_bytes = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in allf)
                out["hbm"] = {"gbytes_per_step_pmc": round(step_bytes / 1e9, 3), "gbytes_per_s": round(step_bytes / elapsed * args.steps / 1e9, 1),
                              "frac_of_peak": round(step_bytes / elapsed * args.steps / 1e9 / HBM_PEAK_GBS, 4), "peak_gbytes_per_s": HBM_PEAK_GBS,
                              "note": "whole step: PMC bytes of one steady step (FETCH_SIZE x 2 + WRITE_SIZE) / this run's ms_per_step"}
            with open(newest("rocprofv3_kernel_summary")) as fh:
                rp = json.load(fh)
            if rp.get("csrc_sha") == sha and rp.get("config") == tag:
                out["roofline"]["avg_launch_us_rocprofv3"] = rp["conv_igemm_all"]["avg_us"]
                out["roofline"]["achieved_rocprofv3"] = round(d["flops"] / (rp["conv_igemm_all"]["us_per_step"] * 1e-6) / 1e12, 2)
            with open(newest("pmc_mfma_util")) as fh:
                mu = json.load(fh)
            if mu.get("csrc_sha") == sha and mu.get("config") == tag:
                out["roofline"]["mfma_busy_frac_pmc"] = mu["conv_igemm_all"]["mfma_util"]
        except (OSError, KeyError, ValueError):
            pass
        # ---- context for `frac` (never a replacement of it): what the matrix pipes of THIS device sustain on random operands.  `peak` is the
        # 2.5 PFLOP/s of 2.4 GHz; under dense MFMA load on real data the chip holds 1.6-1.9 GHz (rtd_bench_mfma_rate: operands in registers,
        # no memory traffic), and the f16x3 arithmetic issues 3 MFMA flops per algorithmic flop.
        if not args.no_mfma_probe and args.precision != "fp32":
            try:
                buf = (C.c_float * 3)()
                rates = {}
                for rnd in (0, 1):
                    rc = _capi.lib().rtd_bench_mfma_rate(rnd, 40, buf)
                    if rc != 0:
                        raise RuntimeError("rtd_bench_mfma_rate")
                    rates["random" if rnd else "zero"] = {"tflops": round(buf[0], 1), "core_clock_ghz": round(buf[1], 3)}
                sus = rates["random"]["tflops"]
                out["roofline"]["sustained_mfma"] = {
                    "operands_zero": rates["zero"], "operands_random": rates["random"],
                    "frac_of_sustained": round(achieved / sus, 4), "mfma_issue_frac_of_sustained": round(mpp * achieved / sus, 4),
                    "note": "v_mfma_f32_16x16x32_f16 back to back on every CU, operands in registers (measured now, this device): the rate the "
                            "matrix pipes hold on random operand bits is the ceiling a kernel on real activations can reach; `frac` stays against the 2.5 PFLOP/s peak"}
            except Exception as e:                                    # a probe failure never costs the bench line
                print(f"[bench] mfma rate probe failed: {e}", file=sys.stderr)
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
                eng.infer(one, 0.25, True, on_device=True)
                if i >= 10:
                    lat.append((time.perf_counter() - t1) * 1e3)
            out["p50_ms_per_frame_bs1"] = round(float(np.percentile(lat, 50)), 4)
            out["p99_ms_per_frame_bs1"] = round(float(np.percentile(lat, 99)), 4)
    for e in engs:
        e.close()

    # ---- several batches in flight (same engine), and the plain-bf16 engine for comparison: extra lines, never `value` ----
    MS = args.multi_streams
    if MS > 1 and MS != S and args.workload == "detect":
        el, es, _ = measure(args.precision, MS)
        out["multi_stream"] = {"value": round(world * B * args.steps / el, 2), "unit": "frames/s", "streams_per_gpu": MS,
                               "ms_per_step": round(1000.0 * el / args.steps, 4),
                               "note": f"same K steps round-robin over {MS} throughput-profile handles ({MS} bs-{B} batches in flight)"}
        for e in es:
            e.close()
    if args.precision == "f16x3" and not args.no_bf16_line and args.workload == "detect" and world == 1:
        el, es, _ = measure("bf16", 1)
        out["bf16_engine"] = {"value": round(world * B * args.steps / el, 2), "unit": "frames/s", "streams_per_gpu": 1,
                              "ms_per_step": round(1000.0 * el / args.steps, 4),
                              "note": "plain bf16 storage + MFMA engine, one batch in flight: outside the 1e-3 / 1e-2 px tolerance (tests/test_gpu_parity.py), shown for comparison"}
        for e in es:
            e.close()

    if rank == 0 and world == 1 and not args.no_detect_host and args.workload == "detect" and args.precision == "f16x3" and not args.opt:
        # ---- the call the product makes: RTDETRDetector.detect(host frame), protocol of the reference's tests/test_inference.py:63-115 ----
        print("[bench] detect() on host frames ...", file=sys.stderr, flush=True)
        import logging
        from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
        logging.getLogger("telescope_cam_detection_amd").setLevel(logging.ERROR)
        dh = {"protocol": "10 warm-ups + 100 timed synchronous RTDETRDetector.detect(np.ndarray HWC uint8 BGR, host memory) per frame size; "
                          "input_size 640x640, conf_threshold 0.25, wildlife_only, precision f16x3, hipGraph; ms per call"}
        for an in ("r50", "r18"):
            det = RTDETRDetector(config_path=an, model_path=f"synthetic:{an}:0", device=f"cuda:{local_rank}", input_size=(640, 640), max_batch=1)
            if not det.load_model(max_retries=1):
                continue
            for (fh_, fw_) in ((640, 640), (720, 1280), (1080, 1920)):
                fr = np.random.default_rng(7000 + fh_).integers(0, 255, (fh_, fw_, 3), dtype=np.uint8)
                for _ in range(10):
                    det.detect(fr)
                ts = []
                for _ in range(100):
                    t1 = time.perf_counter()
                    det.detect(fr)
                    ts.append((time.perf_counter() - t1) * 1e3)
                ts = np.asarray(ts)
                dh[f"{an}_{fw_}x{fh_}"] = {k: round(float(v), 3) for k, v in (("mean", ts.mean()), ("std", ts.std()), ("min", ts.min()), ("max", ts.max()),
                                                                           ("p50", np.percentile(ts, 50)), ("p95", np.percentile(ts, 95)), ("p99", np.percentile(ts, 99)))}
                dh[f"{an}_{fw_}x{fh_}"]["fps"] = round(1000.0 / float(ts.mean()), 1)
            det.model.engine.close()
        out["detect_host_ms"] = dh

    if rank == 0:
        # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
        if not args.no_cpu_baseline and world == 1:
            from oracle import rtdetr_oracle as orc
            # the box's CPU share, not the host's core count (oversubscribed OpenMP spin-waits look like a hang)
            cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("RTD_CPU_THREADS", "16")))
            torch.set_num_threads(cores)
            print(f"[bench] cpu baseline on {cores} threads ...", file=sys.stderr, flush=True)
            nb = min(B, 8)                                    # the benchmark batch itself (R50 640 bs 8: ~2 s per batch on 16 threads)
            host = [f.cpu().numpy() for f in frames[:nb]]
            w = synth_weights(arch, 0)                       # the oracle takes the un-folded state (the engines were loaded from the cached blob)
            orc.detect_batch(arch, w, host, (H, H))          # warm-up
            reps = 3
            t1 = time.perf_counter()
            for _ in range(reps):
                orc.detect_batch(arch, w, host, (H, H))
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(nb * reps / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"CPU oracle (fp32 eager PyTorch restatement of the reference path), RT-DETR-{args.arch.upper()} "
                                             f"{H}x{H} bs={nb}, {reps} timed batches after 1 warm-up, torch threads={cores}"}
            # BASELINE configs[0] by the reference's own protocol (tests/test_inference.py:76-115): R18 640x640, ONE frame per call, 10 warm-ups,
            # then timed single-frame calls with mean/std/min/max/p50/p95/p99 - the CPU figure beside detect_host_ms.r18_640x640
            print("[bench] cpu baseline R18 bs=1 ...", file=sys.stderr, flush=True)
            a18 = ARCHS["r18"]
            w18 = synth_weights(a18, 0)
            fr18 = np.random.default_rng(7640).integers(0, 255, (640, 640, 3), dtype=np.uint8)
            for _ in range(10):
                orc.detect_batch(a18, w18, [fr18], (640, 640))
            ts = []
            for _ in range(30):
                t1 = time.perf_counter()
                orc.detect_batch(a18, w18, [fr18], (640, 640))
                ts.append((time.perf_counter() - t1) * 1e3)
            ts = np.asarray(ts)
            out["cpu_baseline_r18_bs1"] = {
                **{k: round(float(v), 2) for k, v in (("mean", ts.mean()), ("std", ts.std()), ("min", ts.min()), ("max", ts.max()),
                                                      ("p50", np.percentile(ts, 50)), ("p95", np.percentile(ts, 95)), ("p99", np.percentile(ts, 99)))},
                "fps": round(1000.0 / float(ts.mean()), 2), "unit": "ms per frame", "cores": cores, "kind": "port",
                "sample": f"CPU oracle, RT-DETR-R18 640x640, one uint8 BGR frame per call (detect_batch of 1), 10 warm-ups + 30 timed calls, "
                          f"torch threads={cores}; the GPU side of the same protocol is detect_host_ms.r18_640x640"}
        print(json.dumps(out), flush=True)
    if rccl_ranks:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
