import os, sys
sys.path.insert(0, ".")
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.synth import noise_frame
from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights
arch = ARCHS["r50"]; B = 8
eng = _capi.Engine(arch, pack_blob(fold_weights(arch, synth_weights(arch, 0))), 0, _capi.PREC_F16X3, B, (640, 640), use_graph=False)
frames = [noise_frame(i, 640, 640) for i in range(B)]
for _ in range(2): eng.infer_raw(frames)
prof = eng.profile(B, 30)
print(os.environ.get("RTD_LIB_PATH", "new"), [(p["name"], round(p["ms"] * 1e3, 1)) for p in prof if p["name"] in ("dec.vp_all", "backbone.s2.b0.c3", "dec.enc_out.fc")])
