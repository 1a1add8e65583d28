#!/usr/bin/env python3
"""Per-kernel HIP-event profile of one forward (rtd_profile), optionally A/B-ing debug options in ONE process.

    python tools/profile_layers.py --arch r50 --batch 8 --ab split_flex --out gpurun_out/layers.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(prof, title, top=30):
    tot = sum(x["ms"] for x in prof)
    fam = {}
    for x in prof:
        f = fam.setdefault(x["kernel"], [0.0, 0.0, 0.0, 0])
        f[0] += x["ms"]; f[1] += x["flops"]; f[2] += x["bytes"]; f[3] += 1
    print(f"== {title}: {tot:.3f} ms over {len(prof)} launches")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:12s} {v[0]:8.3f} ms n={v[3]:3d}  {v[1] / max(v[0], 1e-9) / 1e9:8.1f} TF/s {v[2] / max(v[0], 1e-9) / 1e6:8.1f} GB/s")
    for x in sorted(prof, key=lambda x: -x["ms"])[:top]:
        print(f"    {x['name']:24s} {x['kernel']:11s} {x['ms'] * 1e3:8.1f} us {x['flops'] / max(x['ms'], 1e-9) / 1e9:8.1f} TF/s "
              f"{x['bytes'] / max(x['ms'], 1e-9) / 1e6:8.1f} GB/s")
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="r50")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--ab", default="", help="debug option to A/B")
    ap.add_argument("--vals", default="1,0", help="two values of the option: baseline,candidate")
    ap.add_argument("--out", default="")
    ap.add_argument("--opt", action="append", default=[], help="name=value set BEFORE the engine is created (plan-build switches such as sc_fold)")
    args = ap.parse_args()
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.weights import fold_weights, pack_blob, synth_weights
    arch = ARCHS[args.arch]
    blob = pack_blob(fold_weights(arch, synth_weights(arch, 0)))
    prec = _capi.precision_code(args.precision)
    for o in args.opt:
        k, v = o.split("=")
        _capi.debug_option(k, int(v))
    res = {}
    VALS = [int(v) for v in args.vals.split(",")]
    engs = {}
    if args.ab:
        # the dispatch switches are snapshotted per handle at rtd_create: one engine per value, profiled in interleaved rounds in ONE process
        # (boxes of the pool differ by several per cent in the clock they hold: never compare two runs)
        for val in VALS:
            _capi.debug_option(args.ab, val)
            engs[val] = _capi.Engine(arch, blob, 0, prec, args.batch, (args.size, args.size), use_graph=False)
        _capi.debug_option("reset", 0)
        eng = engs[VALS[0]]
        for rnd in range(3):
            for val in VALS:
                p = engs[val].profile(args.batch, args.reps)
                res.setdefault(f"{args.ab}={val}", []).append(p)
        for key in list(res):                     # per launch: the fastest of the rounds
            rounds = res[key]
            best = [dict(x) for x in rounds[0]]
            for r in rounds[1:]:
                for bx, x in zip(best, r):
                    bx["ms"] = min(bx["ms"], x["ms"])
            res[key].append(best)
    else:
        eng = _capi.Engine(arch, blob, 0, prec, args.batch, (args.size, args.size), use_graph=False)
    if args.ab:
        a = res[f"{args.ab}={VALS[0]}"][-1]
        b = res[f"{args.ab}={VALS[1]}"][-1]
        ta = summarize(a, f"{args.ab}={VALS[0]}", top=0)
        tb = summarize(b, f"{args.ab}={VALS[1]}", top=40)
        print(f"total {ta:.3f} -> {tb:.3f} ms")
        print(f"per-layer (us)  {args.ab}={VALS[0]} -> {VALS[1]}")
        for x, y in zip(a, b):
            if x["kernel"] == "conv_igemm" and abs(x["ms"] - y["ms"]) > 0.0015:
                print(f"    {x['name']:24s} {x['ms'] * 1e3:8.1f} -> {y['ms'] * 1e3:8.1f}  ({x['ms'] / max(y['ms'], 1e-9):.2f}x)  "
                      f"{y['flops'] / y['ms'] / 1e9:7.1f} TF/s {y['bytes'] / y['ms'] / 1e6:7.1f} GB/s")
    else:
        p = eng.profile(args.batch, args.reps)
        res["profile"] = [p]
        summarize(p, "profile", top=60)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({k: v[-1] for k, v in res.items()}, fh)
    for e in (list(engs.values()) or [eng]):
        e.close()


if __name__ == "__main__":
    main()
