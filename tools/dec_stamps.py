#!/usr/bin/env python3
"""Diagnostic: per-phase time of the fused decoder-layer kernel (s_memrealtime stamps, layer 2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.synth import noise_frame
from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights
arch = ARCHS["r50"]; B = 8
# RTD_DEC_PROBE: 1 = the linear layers skip their MFMAs (filter stream alone), 2 = skip their filter loads (arithmetic alone); results are wrong, timings only
_capi.debug_option("dec_stamps", 1 + 2 * int(os.environ.get("RTD_DEC_PROBE", "0")))
eng = _capi.Engine(arch, pack_blob(fold_weights(arch, synth_weights(arch, 0))), 0, _capi.precision_code(os.environ.get("RTD_PREC", "f16x3")), B, (640, 640), use_graph=False)
frames = [noise_frame(i, 640, 640) for i in range(B)]
for _ in range(3):
    eng.infer_raw(frames)
raw = eng.debug_tensor("dec_stamps")[0, :, 0, :] * 0.01   # us
st = np.concatenate([raw[:, 0:1], raw[:, 12:13], raw[:, 1:12]], 1)
names = ["load", "self-attn", "o_proj", "ln1", "add+offaw", "sampling", "op+ln2", "fc1", "fc2+ln3", "bbox+refine", "hs st+qpos", "qk+v", "stores"]
d = np.diff(np.concatenate([np.zeros((st.shape[0], 1)), st], 1), axis=1)
print("blocks", st.shape[0], " total us: median %.1f  max %.1f" % (np.median(st[:, 12]), st[:, 12].max()))
for i, nme in enumerate(names):
    print(f"  {nme:12s} median {np.median(d[:, i]):7.2f} us   p90 {np.percentile(d[:, i], 90):7.2f}")
eng.close()
