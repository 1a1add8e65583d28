"""Prints rtd_self_check's report for the tinyc network with one stage-2 channel scaled up (tests/test_weights_guard.py's cases)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from telescope_cam_detection_amd import _capi
from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights

arch = ARCHS["tinyc"]
for gamma in [1.0] + [float(v) for v in sys.argv[1:]]:
    w = {k: v.clone() for k, v in synth_weights(arch, 3).items()}
    w["backbone.s2.b0.c1.bn.v"][5] = 1e-6
    w["backbone.s2.b0.c1.bn.g"][5] = gamma
    w["backbone.s2.b0.c1.bn.m"][5] = -1.0                      # the channel's pre-activation is positive everywhere: it survives the ReLU
    blob = pack_blob(fold_weights(arch, w))
    try:
        e = _capi.Engine(arch, blob, device=0, precision=_capi.PREC_F16X3, max_batch=1, input_size=(160, 224))
        print(gamma, e.self_check(blob))
        e.close()
    except Exception as ex:
        print(gamma, "refused:", str(ex)[:200])
