"""How coarse may the decoder's value maps be?  The oracle network with the value projection's OUTPUT rounded to a storage format
(everything else fp32): fp16 (11 bits), bf16x2 / f16x2 pairs, "q3" = hi fp16 + 8-bit lo with a per-(token, head) shared scale.
Prints the worst box / score error of the final rows against the un-rounded oracle.  (Test tooling; EXPERIMENTS.md round 5.)

    python tools/value_fmt_sim.py c2_r50_640_scene_bs2 f16 q3row
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rtdetr_oracle as orc  # noqa: E402
from util import load_case, match_detections, weights_for  # noqa: E402


def fmt(v, kind, heads):
    if kind == "f16":
        return v.half().float()
    if kind == "f16x2":
        hi = v.half().float()
        return hi + (v - hi).half().float()
    if kind == "q3row":      # hi fp16 + int8 lo, one scale per (token, head) = per 32-channel row the sampler reads
        hi = v.half().float()
        lo = v - hi
        shp = v.shape
        lo = lo.view(*shp[:-1], heads, shp[-1] // heads)
        s = lo.abs().amax(-1, keepdim=True).clamp_min(1e-30) / 127.0
        q = torch.round(lo / s).clamp(-127, 127)
        return hi + (q * s).view(shp)
    raise ValueError(kind)


def main():
    case = sys.argv[1]
    kinds = sys.argv[2:] or ["f16", "q3row"]
    torch.set_num_threads(8)
    arch, wseed, size, frames, g = load_case(case)
    w = weights_for(arch, wseed)
    xs, sizes = zip(*[orc.preprocess(f, size) for f in frames])
    xs = torch.cat(xs, 0)
    lin = orc.linear
    with torch.no_grad():
        rl, rb, rs = orc.model_forward(arch, w, xs, list(sizes))
    for kind in kinds:
        def patched(w_, name, x):
            y = lin(w_, name, x)
            return fmt(y, kind, arch.dec_heads) if name.endswith(".vp") else y
        orc.linear = patched
        try:
            with torch.no_grad():
                l, b, s = orc.model_forward(arch, w, xs, list(sizes))
        finally:
            orc.linear = lin
        for i in range(len(frames)):
            m, n, ws, wb = match_detections(rl[i], rb[i], rs[i], l[i], b[i], s[i], 1e-3, 1e-2)
            m2, _, ws2, wb2 = match_detections(rl[i], rb[i], rs[i], l[i], b[i], s[i], 1e-2, 1.0)
            print(f"{kind:6s} frame {i}: matched at 1e-3 / 1e-2 px: {m}/{n} (worst {ws:.1e} / {wb:.1e} px); at 1e-2 / 1 px: {m2}/{n} (worst {ws2:.1e} / {wb2:.1e} px)", flush=True)


if __name__ == "__main__":
    main()
