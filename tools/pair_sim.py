"""CPU simulation of the hi/lo pair formats on the oracle network (test tooling, never on the product path).

Every conv / linear operand (activation and filter) is rounded to hi + lo in the given 16-bit format before the product,
accumulation stays fp32 - the arithmetic of the GPU's pair engines without their kernels.  Prints the worst box / score
error of the final rows against the un-rounded fp32 oracle, to choose the storage format for large inputs:

    python tools/pair_sim.py c3_r101_1280_bs1 bf16 f16
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import rtdetr_oracle as orc  # noqa: E402
from util import load_case, match_detections, weights_for  # noqa: E402


FTZ = False   # "f16ftz": what the pair would keep if the matrix cores flushed fp16 subnormal inputs


def pair(x, dt):
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    if FTZ:
        hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi + lo


def run(arch, w, xs, sizes, dt):
    c2, li = F.conv2d, F.linear
    if dt is not None:
        F.conv2d = lambda x, w_, b=None, **kw: c2(pair(x, dt), pair(w_, dt), b, **kw)
        F.linear = lambda x, w_, b=None: li(pair(x, dt), pair(w_, dt), b)
    try:
        with torch.no_grad():
            return orc.model_forward(arch, w, xs, sizes)
    finally:
        F.conv2d, F.linear = c2, li


def main():
    case = sys.argv[1]
    fmts = sys.argv[2:] or ["bf16", "f16"]
    torch.set_num_threads(8)
    arch, wseed, size, frames, g = load_case(case)
    w = weights_for(arch, wseed)
    xs, sizes = zip(*[orc.preprocess(f, size) for f in frames])
    xs = torch.cat(xs, 0)
    t = time.time()
    rl, rb, rs = run(arch, w, xs, list(sizes), None)
    print(f"fp32 oracle {time.time() - t:.1f} s", flush=True)
    for f in fmts:
        global FTZ
        FTZ = f == "f16ftz"
        dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f16ftz": torch.float16}[f]
        l, b, s = run(arch, w, xs, list(sizes), dt)
        for i in range(len(frames)):
            m, n, ws, wb, un = match_detections(rl[i], rb[i], rs[i], l[i], b[i], s[i], 1e-3, 1e9, return_unmatched=True)
            # box error of every matched row (label + score matched), sorted
            d = []
            used = np.zeros(len(l[i]), bool)
            for r in range(len(rl[i])):
                c = np.where((l[i].numpy() == int(rl[i][r])) & ~used & (np.abs(s[i].numpy() - float(rs[i][r])) <= 1e-3))[0]
                if len(c) == 0:
                    continue
                e = np.abs(b[i].numpy()[c] - rb[i].numpy()[r]).max(axis=1)
                j = int(np.argmin(e)); used[c[j]] = True; d.append(e[j])
            d = np.sort(np.array(d))[::-1]
            print(f"{f} frame {i}: score-matched {m}/{n}  worst dbox px {d[:5]}  rows > 1e-2 px: {(d > 1e-2).sum()}  dscore {ws:.2e}", flush=True)


if __name__ == "__main__":
    main()
