#!/usr/bin/env python3
"""Diagnostic: one golden case on the f16x3 engine with rtd_debug_option settings from the command line; prints matched rows per tolerance.
    python tools/x3_case.py c3_r101_1280_bs1 conv_reg=1 split_sx=0"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.weights import fold_weights, pack_blob
from tests.util import load_case, match_detections, weights_for
name = sys.argv[1]
for o in sys.argv[2:]:
    k, v = o.split("=")
    _capi.debug_option(k, int(v))
arch, wseed, input_size, frames, g = load_case(name)
eng = _capi.Engine(arch, pack_blob(fold_weights(arch, weights_for(arch, wseed))), 0, _capi.precision_code(os.environ.get("RTD_PREC", "f16x3")), len(frames), input_size, use_graph=False)
labels, boxes, scores = eng.infer_raw(frames)
mx = eng.debug_tensor("enc_cls_max")[:, :, 0, 0]
print(name, sys.argv[2:], "enc score max abs err %.2e" % np.abs(mx - g["enc_cls_max"]).max())
for b in range(len(frames)):
    d = []
    for tol in (5e-3, 1e-2, 2e-2):
        m, n, ws, wb = match_detections(g["labels"][b], g["boxes"][b], g["scores"][b], labels[b], boxes[b], scores[b], 1e-3, tol)
        d.append(f"{tol:g}px: {m}/{n}")
    m, n, ws, wb = match_detections(g["labels"][b], g["boxes"][b], g["scores"][b], labels[b], boxes[b], scores[b], 1e-3, 1.0)
    print(f"  [{b}] " + "  ".join(d) + f"   worst dscore {ws:.2e} dbox {wb:.2e}")
eng.close()
