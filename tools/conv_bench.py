#!/usr/bin/env python3
"""Isolated conv-layer timings (rtd_bench_conv): warm operands vs operands flushed to HBM, per debug option.

    python tools/conv_bench.py [--opt conv_mode --vals 0,6]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, H=W, Cin, Cout, k, stride, pad, with_res)  -- R50 640x640 bs8 layers
SHAPES = [
    ("s0.c1 1x1 256->64 @160", 160, 256, 64, 1, 1, 0, 0),
    ("s0.c2 3x3 64->64 @160", 160, 64, 64, 3, 1, 1, 0),
    ("s0.c3 1x1 64->256 @160 +res", 160, 64, 256, 1, 1, 0, 1),
    ("s1.c2 3x3 128->128 @80", 80, 128, 128, 3, 1, 1, 0),
    ("s1.c3 1x1 128->512 @80 +res", 80, 128, 512, 1, 1, 0, 1),
    ("s2.c1 1x1 1024->256 @40", 40, 1024, 256, 1, 1, 0, 0),
    ("s2.c2 3x3 256->256 @40", 40, 256, 256, 3, 1, 1, 0),
    ("s2.c3 1x1 256->1024 @40 +res", 40, 256, 1024, 1, 1, 0, 1),
    ("s3.c1 1x1 2048->512 @20", 20, 2048, 512, 1, 1, 0, 0),
    ("s3.c2 3x3 512->512 @20", 20, 512, 512, 3, 1, 1, 0),
    ("s3.c3 1x1 512->2048 @20 +res", 20, 512, 2048, 1, 1, 0, 1),
    ("fpn1.c12 1x1 512->512 @80", 80, 512, 512, 1, 1, 0, 0),
    ("proj.0 1x1 512->256 @80", 80, 512, 256, 1, 1, 0, 0),
    ("fpn1.rep 3x3 256->256 @80", 80, 256, 256, 3, 1, 1, 0),
    ("pan1.rep 3x3 256->256 @20", 20, 256, 256, 3, 1, 1, 0),
    ("lat 1x1 256->256 @20", 20, 256, 256, 1, 1, 0, 0),
    ("vp_all 1x1 256->1536 @8400tok", 0, 256, 1536, 1, 1, 0, 0),
    # per-block fixed cost: 1x1, 80^2 x 8 pixels, 256 output channels (800 tiles), K = 32 .. 2048 (1 .. 64 K-steps of the split kernels)
    ("kfix 1x1 32->256 @80", 80, 32, 256, 1, 1, 0, 0),
    ("kfix 1x1 64->256 @80", 80, 64, 256, 1, 1, 0, 0),
    ("kfix 1x1 128->256 @80", 80, 128, 256, 1, 1, 0, 0),
    ("kfix 1x1 256->256 @80", 80, 256, 256, 1, 1, 0, 0),
    ("kfix 1x1 512->256 @80", 80, 512, 256, 1, 1, 0, 0),
    ("kfix 1x1 1024->256 @80", 80, 1024, 256, 1, 1, 0, 0),
    ("kfix 1x1 2048->256 @80", 80, 2048, 256, 1, 1, 0, 0),
    # tile-count quantization probes: 3x3 256->256 with 512 / 648 / 800 / 1024 tiles of 128 x 128
    ("quant 3x3 256->256 @64", 64, 256, 256, 3, 1, 1, 0),
    ("quant 3x3 256->256 @72", 72, 256, 256, 3, 1, 1, 0),
    ("quant 3x3 256->256 @80", 80, 256, 256, 3, 1, 1, 0),
    ("quant 3x3 256->256 @88", 88, 256, 256, 3, 1, 1, 0),
    ("quant 3x3 256->256 @90", 90, 256, 256, 3, 1, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", default="")
    ap.add_argument("--vals", default="0")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--flush-mb", type=int, default=1024)
    ap.add_argument("--dtype", type=int, default=4, help="0 bf16, 1 fp32, 4 F16X2 (the f16x3 engine's operands)")
    ap.add_argument("--only", default="", help="substring filter on the layer name")
    ap.add_argument("--set", action="append", default=[], help="name=value: further debug options held fixed (e.g. bench_rewarm=32: random operands)")
    args = ap.parse_args()
    from telescope_cam_detection_amd import _capi
    L = _capi.lib()
    vals = [int(v) for v in args.vals.split(",")]
    for kv in args.set:
        n, v = kv.split("=")
        _capi.debug_option(n, int(v))
    print(f"{'layer':34s} " + "  ".join(f"{args.opt or 'default'}={v}: warm / cold us" for v in vals))
    for name, hw, cin, cout, k, st, pad, res in SHAPES:
        if args.only and args.only not in name:
            continue
        row = []
        for v in vals:
            if args.opt:
                _capi.debug_option(args.opt, v)
            out = (C.c_float * 2)()
            if hw == 0:      # token GEMM: 8400 tokens per image as a 1 x 8400 "image"
                rc = L.rtd_bench_conv(args.dtype, args.batch, 1, 8400, cin, cout, k, st, pad, res, args.reps, args.flush_mb, out)
            else:
                rc = L.rtd_bench_conv(args.dtype, args.batch, hw, hw, cin, cout, k, st, pad, res, args.reps, args.flush_mb, out)
            assert rc == 0, _capi.last_error() if hasattr(_capi, "last_error") else rc
            row.append(f"{out[0]:7.1f} / {out[1]:7.1f}")
        print(f"{name:34s} " + "    ".join(row), flush=True)
    if args.opt:
        _capi.debug_option(args.opt, 0)


if __name__ == "__main__":
    main()
