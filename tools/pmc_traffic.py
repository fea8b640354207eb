#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) into per-launch HBM traffic per kernel.  gfx950 corrections (MI355X_MICROARCH.md, HBM):
FETCH_SIZE is reported in KiB and counts 64 B per 128-B request on wide coalesced reads -> x2;
WRITE_SIZE in KiB is exact for streaming stores.
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <out.md> [name of the committed .md] [profiled command]"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(ltxmi::(?:\w+::)*\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def collect(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k:
            acc[(k, r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    rows = []
    for key in sorted(set(fetch) | set(write)):
        f = fetch.get(key, [])
        w = write.get(key, [])
        fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wb = 1024.0 * sum(w) / max(len(w), 1)
        rows.append({"kernel": key[0], "grid": key[1], "launches": max(len(f), len(w)),
                     "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                     "hbm_bytes_per_launch": fb + wb})
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
    out = {"rows": rows}
    # the bench's dominant kernel: FF up-projection (EPI 1 = GELU) on the persistent 256x256 kernel
    for r in rows:
        if "persistent" in r["kernel"] and r["kernel"].rstrip(">").endswith(", 1"):
            out["ff1_gemm_hbm_bytes_per_launch"] = r["hbm_bytes_per_launch"]
            break
    # the bench's roofline kernel: pipelined self-attention (the launch with the largest grid = the DiT's self-attention)
    att = [r for r in rows if "attn_pipe_kernel" in r["kernel"] or "attn_pipe_persistent_kernel" in r["kernel"]]
    if att:
        out["attention_hbm_bytes_per_launch"] = max(att, key=lambda r: r["launches"])["hbm_bytes_per_launch"]
    # the VAE legs' roofline kernel: the direct convolution with the plain store and the largest grid (128 -> 128 at the
    # full-resolution stage)
    # (conv1 of a ResnetBlock3D there: epilogue 3 = bias + norm2 -> SiLU; the plain-store instance is the fallback)
    cv = ([r for r in rows if "conv3d_direct_v3_kernel<3>" in r["kernel"]] or [r for r in rows if "conv3d_direct_v3_kernel<0>" in r["kernel"]]
          or [r for r in rows if "conv3d_direct_kernel<0>" in r["kernel"]])
    if cv:
        out["conv_direct_hbm_bytes_per_launch"] = max(cv, key=lambda r: int(r["grid"] or 0))["hbm_bytes_per_launch"]
    out["source"] = sys.argv[5] if len(sys.argv) > 5 else sys.argv[4]
    what = sys.argv[6] if len(sys.argv) > 6 else "python bench.py --steps 1 --warmup 1 --no-extras"
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    with open(sys.argv[4], "w") as f:
        f.write("# HBM traffic per launch from PMC counters\n\n"
                "`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes over "
                f"`{what}`; FETCH_SIZE x2 (gfx950 counts 64 B per "
                "128-B request), both KiB -> bytes.\n\n| kernel | grid | launches | fetch MB | write MB | total MB |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write(f"| `{r['kernel']}` | {r['grid']} | {r['launches']} | {r['fetch_bytes_per_launch'] / 1e6:.1f} | "
                    f"{r['write_bytes_per_launch'] / 1e6:.1f} | {r['hbm_bytes_per_launch'] / 1e6:.1f} |\n")


if __name__ == "__main__":
    main()
