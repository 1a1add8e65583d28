"""Same-box A/B of one rtd_debug_option on the fused decoder / AIFI launches (R50 bs 8): per-layer HIP-event times and whether the raw
outputs stay bit-identical.    python tools/dec_ab.py attn_split 2 3"""
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
opt = sys.argv[1] if len(sys.argv) > 1 else "attn_split"
v0, v1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 3)
outs = {}
for flat in (v0, v1, v0, v1):
    _capi.debug_option(opt, flat)
    eng = _capi.Engine(arch, blob, 0, _capi.PREC_F16X3, B, (640, 640), use_graph=False)
    for _ in range(3):
        o = eng.infer_raw(frames)
    prof = eng.profile(B, 20)
    dec = [(p["name"], p["ms"] * 1e3) for p in prof if p["kernel"] == "dec_layer"]
    print(f"{opt}={flat}: dec_layer total {sum(v for _, v in dec):7.1f} us  " + " ".join(f"{n.split('.')[-2] if n.count('.') > 1 else n}:{v:.1f}" for n, v in dec), flush=True)
    if flat in outs:
        pass
    outs.setdefault(flat, o)
    eng.close()
same = all(np.array_equal(x, y) for x, y in zip(outs[v0], outs[v1]))
print(f"outputs bit-identical between {opt} {v0} and {v1}:", same)
if not same:
    for x, y in zip(outs[v0], outs[v1]):
        d = np.abs(x.astype(np.float64) - y.astype(np.float64))
        print("  max abs diff", d.max())
