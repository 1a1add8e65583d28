#!/usr/bin/env python3
"""Do two conv launches on two HIP streams overlap?  (rtd_bench_conv_pair)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi

L = _capi.lib()
SH = {  # B, HW, Cin, Cout, K, stride, pad
    "s3c2 (ws4,100 blk)": (8, 20, 512, 512, 3, 1, 1),
    "s2c2 (ws4,200 blk)": (8, 40, 256, 256, 3, 1, 1),
    "s1c2 (ws4,400 blk)": (8, 80, 128, 128, 3, 1, 1),
    "fpn1 (ws2,800 blk)": (8, 80, 256, 256, 3, 1, 1),
    "s0c3 (ws2,3200 blk)": (8, 160, 64, 256, 1, 1, 0),
    "s0c1 (v1,1600 blk)": (8, 160, 256, 64, 1, 1, 0),
    "s0c2 (reg,256 blk)": (8, 160, 64, 64, 3, 1, 1),
}
pairs = [("s3c2 (ws4,100 blk)", "s3c2 (ws4,100 blk)"), ("s3c2 (ws4,100 blk)", "s2c2 (ws4,200 blk)"), ("s2c2 (ws4,200 blk)", "s2c2 (ws4,200 blk)"),
         ("s3c2 (ws4,100 blk)", "fpn1 (ws2,800 blk)"), ("s3c2 (ws4,100 blk)", "s0c3 (ws2,3200 blk)"), ("s3c2 (ws4,100 blk)", "s0c1 (v1,1600 blk)"),
         ("s3c2 (ws4,100 blk)", "s0c2 (reg,256 blk)"), ("s1c2 (ws4,400 blk)", "s1c2 (ws4,400 blk)"), ("fpn1 (ws2,800 blk)", "fpn1 (ws2,800 blk)"),
         ("s0c3 (ws2,3200 blk)", "s0c3 (ws2,3200 blk)"), ("s0c1 (v1,1600 blk)", "s0c2 (reg,256 blk)")]
for a, b in pairs:
    out = (C.c_float * 3)()
    sa = (C.c_int32 * 7)(*SH[a]); sb = (C.c_int32 * 7)(*SH[b])
    rc = L.rtd_bench_conv_pair(sa, sb, 50, out)
    assert rc == 0
    print(f"{a:22s} + {b:22s}: alone {out[0]:7.1f} / {out[1]:7.1f} us, together {out[2]:7.1f} us  (serial {out[0] + out[1]:7.1f}, overlap gain {(out[0] + out[1]) / out[2]:.2f}x)", flush=True)
