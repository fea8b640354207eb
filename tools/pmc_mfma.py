#!/usr/bin/env python3
"""Per-kernel matrix-pipe / VALU / wait fractions from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE).
Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip,
GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.
  mfma_busy   = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
  valu / wait = ACTIVE_INST_VALU, WAIT_ANY, WAIT_INST_ANY as fractions of WAVE_CYCLES
Usage: pmc_mfma.py <counter_collection.csv> <out.md>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(ltxmi::(?:\w+::)*\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        if k:
            acc[(k, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for key, c in acc.items():
        mean = {n: sum(v) / len(v) for n, v in c.items()}
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
        wave = mean.get("SQ_WAVE_CYCLES", 0.0)
        if gui <= 0 or wave <= 0:
            continue
        rows.append((gui * len(c["GRBM_GUI_ACTIVE"]), key[0], key[1], len(c["GRBM_GUI_ACTIVE"]), gui / 8,
                     mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024),
                     mean.get("SQ_ACTIVE_INST_VALU", 0.0) / wave, mean.get("SQ_WAIT_ANY", 0.0) / wave,
                     mean.get("SQ_WAIT_INST_ANY", 0.0) / wave))
    rows.sort(reverse=True)
    with open(sys.argv[2], "w") as f:
        f.write("# Matrix-pipe utilisation per kernel from PMC counters \n\n"
                "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                "GRBM_GUI_ACTIVE -- python bench.py --steps 1 --warmup 1 --no-extras`; formulas in tools/pmc_mfma.py. "
                "mfma busy = share of SIMD cycles with the matrix pipe executing (at the clock the chip held).\n\n"
                "| kernel | grid | launches | kernel cycles | mfma busy | VALU-issue / wave-cycles | parked (s_waitcnt, barrier) | issue-stalled |\n"
                "|---|---|---|---|---|---|---|---|\n")
        for _, k, g, n, cyc, mf, va, wa, wi in rows[:16]:
            f.write(f"| `{k}` | {g} | {n} | {cyc:,.0f} | {mf:.1%} | {va:.1%} | {wa:.1%} | {wi:.1%} |\n")


if __name__ == "__main__":
    main()
