#!/usr/bin/env python3
"""In-kernel s_memtime stamps of one conv layer (rtd_bench_conv with rtd_debug_option("glds_drop", 32)).

    RTD_CONV_STAMPS=2 python tools/conv_stamps.py s0c3          # block-level phases of the 128-pixel ws kernels (bf16 operands)
(round 5: the 256-pixel and A-stationary bf16 kernels and their stamp modes were removed with those kernels)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi

SHAPES = {  # HW, Cin, Cout, k, stride, pad, residual
    "s0c3": (160, 64, 256, 1, 1, 0, 1), "s1c3": (80, 128, 512, 1, 1, 0, 1), "s0sc": (160, 64, 256, 1, 1, 0, 0),
    "fpn1": (80, 256, 256, 3, 1, 1, 0), "s1c1": (80, 512, 128, 1, 1, 0, 0), "s1c2": (80, 128, 128, 3, 1, 1, 0),
    "c12": (80, 512, 512, 1, 1, 0, 0), "proj0": (80, 512, 256, 1, 1, 0, 0), "s2c2": (40, 256, 256, 3, 1, 1, 0),
    "lat": (20, 256, 256, 1, 1, 0, 0), "s3c2": (20, 512, 512, 3, 1, 1, 0), "vpall": (-8400, 256, 1536, 1, 1, 0, 0),
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "s1c2"
    L = _capi.lib()
    _capi.debug_option("glds_drop", 32)
    hw, cin, cout, k, st, pad, res = SHAPES[which]
    out = (C.c_float * 2)()
    if hw < 0:      # token GEMM: 1 x (-hw) "image"
        rc = L.rtd_bench_conv(0, 8, 1, -hw, cin, cout, k, st, pad, res, 3, 0, out)
    else:
        rc = L.rtd_bench_conv(0, 8, hw, hw, cin, cout, k, st, pad, res, 3, 0, out)
    print(which, "warm us per launch", round(out[0], 2), "rc", rc)


if __name__ == "__main__":
    main()
