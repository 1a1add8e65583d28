"""How much detection accuracy does the bf16 engine's decoder arithmetic cost?  Runs the fp32 engine (reference) and the bf16
engine under several `dec_split` settings on the same frames / weights and matches the detections frame by frame.

    python tools/dec_precision.py --arch r50 --batch 4
dec_split: 1 = hi/lo splits of both operands (~fp32 products), 2 = bf16 filters x split activations (half the filter bytes).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="r50")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--modes", default="1,2")
    args = ap.parse_args()
    from util import match_detections
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import noise_frame, scene_frame
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights
    arch = ARCHS[args.arch]
    blob = pack_blob(fold_weights(arch, synth_weights(arch, 0)))
    H = args.size
    frames = [scene_frame(300 + i, H, H) if i % 2 else noise_frame(300 + i, H, H) for i in range(args.batch)]

    def run(prec, split):
        _capi.debug_option("dec_split", split)
        e = _capi.Engine(arch, blob, 0, prec, args.batch, (H, H), use_graph=False)
        out = e.infer_raw(frames)
        prof = e.profile(args.batch, 5)
        dec = sum(p["ms"] for p in prof if p["kernel"] == "dec_layer")
        e.close()
        return out, dec

    ref, _ = run(_capi.PREC_FP32, 1)
    for m in [int(v) for v in args.modes.split(",")]:
        (l, b, s), dec = run(_capi.PREC_BF16, m)
        tm = tn = 0
        ws = wb = 0.0
        top_m = top_n = 0
        for i in range(args.batch):
            for tol_s, tol_b, tag in ((2e-2, 2.0, "all"),):
                mm, nn, w1, w2 = match_detections(ref[0][i], ref[1][i], ref[2][i], l[i], b[i], s[i], tol_s, tol_b)
                tm += mm; tn += nn; ws = max(ws, w1); wb = max(wb, w2)
            k = 30                                        # the 30 best-scoring reference detections
            mm, nn, _, _ = match_detections(ref[0][i][:k], ref[1][i][:k], ref[2][i][:k], l[i], b[i], s[i], 1e-2, 1.0)
            top_m += mm; top_n += nn
        # mean |dscore| of same-rank rows (coarse)
        ds = float(np.mean(np.abs(np.asarray(ref[2]) - np.asarray(s))))
        print(f"dec_split={m}: dec_layer {dec * 1e3:7.1f} us/step   matched {tm}/{tn} (2e-2, 2 px)  top-30 {top_m}/{top_n} (1e-2, 1 px)  "
              f"worst dscore {ws:.2e} dbox {wb:.2e}px  mean |dscore by rank| {ds:.2e}")
    _capi.debug_option("dec_split", 1)


if __name__ == "__main__":
    main()
