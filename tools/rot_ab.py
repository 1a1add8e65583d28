"""decoder experiment driver: dec_stamps values -> dec_layer time (R50 bs 8) and bit-equality of the outputs with the default"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.synth import noise_frame, scene_frame
from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights
arch = ARCHS["r50"]; B = 8
blob = pack_blob(fold_weights(arch, synth_weights(arch, 0)))
frames = [scene_frame(i, 640, 640) if i % 2 else noise_frame(i, 640, 640) for i in range(B)]
vals = [int(v) for v in sys.argv[1:]] or [0, 8]
ref = None
for rep in range(2):
    for m in vals:
        _capi.debug_option("dec_stamps", m)
        eng = _capi.Engine(arch, blob, 0, _capi.PREC_F16X3, B, (640, 640), use_graph=False)
        for _ in range(3):
            o = eng.infer_raw(frames)
        if ref is None:
            ref = o
        same = all(np.array_equal(x, y) for x, y in zip(ref, o))
        prof = eng.profile(B, 20)
        dec = [p["ms"] * 1e3 for p in prof if p["kernel"] == "dec_layer"]
        print(f"dec_stamps {m:3d}: dec_layer total {sum(dec):7.1f} us  l2 {dec[5]:.1f}  same as first: {same}", flush=True)
        eng.close()
