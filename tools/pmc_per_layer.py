#!/usr/bin/env python3
"""HBM traffic per conv launch of one steady step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/refresh_profiles.sh) next to the
plan's algorithmic bytes (profiles/*_layers_*.json): which layers re-read their operands from beyond L2.

    python tools/pmc_per_layer.py gpurun_out/r02/pmc_fetch/runc_counter_collection.csv gpurun_out/r02/pmc_write/runc_counter_collection.csv \
        profiles/r02_layers_r50_bs8_f16x3.json

Launches are matched to layers in order: the layer list must come from the same plan as the counter passes (refresh_profiles.sh profiles the
one-stream plan, `--opt side_stream=0`, whose decoder input projections sit after the PAN path: `tools/profile_layers.py --opt side_stream=0`).
"""
import csv
import json
import sys


def load(path, cname):
    rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path)) if r.get("Counter_Name") == cname]
    rows.sort()
    start = [i for i, (_, k, _) in enumerate(rows) if "set_frame_table" in k or "k_preprocess_identity" in k]
    return rows[start[-1]:]                       # the last step (hipGraph replay)


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    layers = [l for l in json.load(open(sys.argv[3]))["profile"] if l["kernel"] == "conv_igemm"]
    ci, tot_a, tot_p = 0, 0.0, 0.0
    for (_, k, f), (_, _, w) in zip(fetch, write):
        if "conv" not in k and "stem0" not in k:
            continue
        if ci >= len(layers):
            break
        l = layers[ci]
        ci += 1
        pmc = (2 * f + w) * 1024                  # FETCH_SIZE x 2: gfx950 correction (MI355X_MICROARCH.md); KiB -> bytes
        tot_a += l["bytes"]
        tot_p += pmc
        flag = "  <<<" if pmc > 1.4 * l["bytes"] and pmc - l["bytes"] > 20e6 else ""
        print(f"{l['name']:24s} {k[10:44]:34s} alg {l['bytes'] / 1e6:7.1f} MB  pmc {pmc / 1e6:7.1f} MB (fetch {2 * f * 1024 / 1e6:6.1f} write {w * 1024 / 1e6:6.1f})"
              f"  x{pmc / l['bytes']:4.2f}{flag}")
    print(f"conv launches matched: {ci}; algorithmic {tot_a / 1e9:.2f} GB, PMC {tot_p / 1e9:.2f} GB per step (x{tot_p / tot_a:.2f})")


if __name__ == "__main__":
    main()
