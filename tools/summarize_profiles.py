#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel trace / PMC counter collection of a bench.py run) into the small JSON summaries
committed under profiles/.

    python tools/summarize_profiles.py trace  gpurun_out/rp/runc/<pid>_kernel_trace.csv        profiles/rNN_rocprofv3_kernel_summary.json
    python tools/summarize_profiles.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> profiles/rNN_pmc_hbm_traffic.json
    python tools/summarize_profiles.py mfma   <counter_collection.csv of --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> profiles/rNN_pmc_mfma_util.json
"""
import collections
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha() -> str:
    """hash of the kernel sources the profile was measured on (bench.py attaches a profile's figures only when it matches)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "telescope_cam_detection_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


CONFIG = os.environ.get("RTD_PROFILE_CONFIG", "r50_640_bs8_f16x3")   # <arch>_<size>_bs<B>_<precision> of the profiled bench.py command
BENCH_ARGS = os.environ.get("RTD_PROFILE_ARGS", "")                  # the extra bench.py arguments of that command (e.g. --arch r101 --size 1280 --batch 4)


def family(name: str) -> str:
    if "conv3x3_reg" in name:
        return "conv3x3_direct"
    if "stem0_u8" in name:
        return "conv_stem0_u8"
    if "conv1x1_sx" in name or "conv1x1_stream" in name:
        return "conv1x1_streaming"
    if "conv_igemm_ws" in name or "conv_igemm_glds" in name:
        return "conv_igemm_lds_dma"
    if "conv_igemm" in name:
        return "conv_igemm_v1"
    m = re.match(r"_ZN3rtd\d+([a-zA-Z_0-9]+?)I", name)
    if m:
        return m.group(1)
    name = re.sub(r"^void ", "", name).replace("rtd::", "")
    return re.sub(r"[<(].*", "", name)


def steady_steps(rows):
    # a step starts with the per-call preprocess launch (outside the hipGraph): the frame-table setter of the fused uint8 stem, else the
    # stand-alone preprocess
    idx = [i for i, r in enumerate(rows) if "k_set_frame_table" in r["Kernel_Name"] or "k_preprocess_identity" in r["Kernel_Name"]]
    steps = [rows[idx[i]:idx[i + 1]] for i in range(len(idx) - 1)]
    modal = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    good = [s for s in steps if len(s) == modal]
    return (good[2:] if len(good) > 4 else good[-1:]), modal


def do_trace(path, out):
    rows = list(csv.DictReader(open(path)))
    steps, modal = steady_steps(rows)
    fam = collections.defaultdict(lambda: [0, 0])
    span = []
    for st in steps:
        span.append((max(int(r["End_Timestamp"]) for r in st) - int(st[0]["Start_Timestamp"])) / 1e6)
        for r in st:
            f = fam[family(r["Kernel_Name"])]
            f[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            f[1] += 1
    tot = sum(v[0] for v in fam.values())
    ks = [dict(kernel=k, us_per_step=round(d / len(steps) / 1e3, 2), launches_per_step=c / len(steps), avg_us=round(d / c / 1e3, 3),
               pct=round(100 * d / tot, 2)) for k, (d, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])]
    conv = [k for k in ks if k["kernel"].startswith("conv")]
    res = dict(csrc_sha=csrc_sha(), config=CONFIG,
               command=f"rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py {BENCH_ARGS} --steps 20 --warmup 3 --streams 1 --multi-streams 0 --no-bf16-line --no-cpu-baseline --no-latency --no-detect-host --no-mfma-probe --opt side_stream=0".replace("  ", " "),
               steady_graph_steps_used=len(steps), kernels_per_step=modal, step_span_ms_median=sorted(span)[len(span) // 2],
               conv_igemm_all=dict(us_per_step=round(sum(k["us_per_step"] for k in conv), 2), launches_per_step=sum(k["launches_per_step"] for k in conv),
                                   avg_us=round(sum(k["us_per_step"] for k in conv) / sum(k["launches_per_step"] for k in conv), 3)),
               kernels=ks)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["conv_igemm_all"]), "step", res["step_span_ms_median"], "ms")
    for k in ks[:8]:
        print(k)


def do_pmc(fetch, write, out):
    def load(path):
        rows = list(csv.DictReader(open(path)))
        steps, _ = steady_steps(rows)
        fam = collections.defaultdict(lambda: [0.0, 0])
        for r in steps[-1]:
            f = fam[family(r["Kernel_Name"])]
            f[0] += float(r["Counter_Value"])
            f[1] += 1
        return fam
    f, w = load(fetch), load(write)
    res = {"csrc_sha": csrc_sha(), "config": CONFIG,
           "note": "one steady graph step; FETCH_SIZE / WRITE_SIZE in KiB from separate rocprofv3 --pmc passes; FETCH_SIZE doubled "
                   "(gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM section)"}
    tot = 0.0
    for k in f:
        fk, c = f[k]
        wk = w[k][0]
        res[k] = dict(launches=c, fetch_kib_raw=fk, fetch_bytes_corrected=fk * 1024 * 2, write_bytes=wk * 1024,
                      hbm_bytes_per_launch=(fk * 2048 + wk * 1024) / c)
        tot += fk * 2048 + wk * 1024
    res["whole_step_hbm_gbytes"] = round(tot / 1e9, 3)
    json.dump(res, open(out, "w"), indent=1)
    print("whole step", res["whole_step_hbm_gbytes"], "GB")
    for k, v in res.items():
        if isinstance(v, dict):
            print(k, v["launches"], round(v["hbm_bytes_per_launch"] / 1e6, 2), "MB/launch")


def do_mfma(path, out):
    """MFMA-pipe utilisation as the counters report it: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over every SIMD of the chip; 32 per
    v_mfma_f32_32x32x16_bf16) over GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs; = shader cycles of the dispatch) x 256 CUs x 4
    SIMDs.  Clock-independent: a kernel whose CUs hold a low clock under MFMA load still shows its pipe occupancy."""
    rows = list(csv.DictReader(open(path)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(r)
    busy_rows = by["SQ_VALU_MFMA_BUSY_CYCLES"]
    act_rows = by["GRBM_GUI_ACTIVE"]
    steps_b, _ = steady_steps(busy_rows)
    steps_a, _ = steady_steps(act_rows)
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
    for rb, ra in zip(steps_b[-1], steps_a[-1]):
        assert rb["Dispatch_Id"] == ra["Dispatch_Id"]
        f = fam[family(rb["Kernel_Name"])]
        f[0] += float(rb["Counter_Value"])
        f[1] += float(ra["Counter_Value"]) / 8.0
        f[2] += 1
        f[3] += (int(rb["End_Timestamp"]) - int(rb["Start_Timestamp"])) / 1e3
    res = {"csrc_sha": csrc_sha(), "config": CONFIG,
           "note": "one steady graph step of `bench.py --streams 1` under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; "
                   "util = busy / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration (reads high on short dispatches)"}
    tb = ta = 0.0
    for k, (b, a, c, us) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        res[k] = dict(launches=c, mfma_busy_cycles=b, shader_cycles=a, us=round(us, 1), mfma_util=round(b / (a * 1024.0), 4) if a else 0.0,
                      mean_clock_ghz=round(a / us / 1e3, 3) if us else 0.0)
        tb += b; ta += a
    res["whole_step"] = dict(mfma_util=round(tb / (ta * 1024.0), 4))
    conv = [v for k, v in res.items() if isinstance(v, dict) and k.startswith("conv")]
    res["conv_igemm_all"] = dict(mfma_util=round(sum(v["mfma_busy_cycles"] for v in conv) / (sum(v["shader_cycles"] for v in conv) * 1024.0), 4))
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if isinstance(v, dict):
            print(k, v)


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        do_trace(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "mfma":
        do_mfma(sys.argv[2], sys.argv[3])
    else:
        do_pmc(sys.argv[2], sys.argv[3], sys.argv[4])
