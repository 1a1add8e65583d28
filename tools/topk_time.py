"""per-step time of the two top-k launches (and the whole tail) from the engine's HIP-event profile, R50 640 bs 8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.synth import noise_frame
from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights
arch = ARCHS[sys.argv[1] if len(sys.argv) > 1 else "r50"]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 8; H = int(sys.argv[3]) if len(sys.argv) > 3 else 640
eng = _capi.Engine(arch, pack_blob(fold_weights(arch, synth_weights(arch, 0))), 0, _capi.PREC_F16X3, B, (H, H), use_graph=False)
frames = [noise_frame(i, H, H) for i in range(B)]
for _ in range(2):
    eng.infer_raw(frames)
prof = eng.profile(B, 30)
for p in prof:
    if p["kernel"] in ("topk", "select_score") or p["name"].startswith(("post.", "dec.enc_topk", "dec.gather")):
        print(f"{p['name']:24s} {p['kernel']:14s} {p['ms'] * 1e3:7.1f} us")
print("total", round(sum(p["ms"] for p in prof) * 1e3, 1), "us")
