#!/bin/bash
# Which kernels the vendor GEMM (hipBLASLt / Tensile via torch.matmul) picks for the hot shapes of a denoise step, and
# how long each takes: rocprofv3 kernel trace of tools/blas_calib.py.  The Tensile solution NAME encodes macro-tile, MFMA
# shape, direct-to-VGPR / direct-to-LDS operands, prefetch depth, LDS buffering, workgroup mapping and stream-K.
#   gpurun -- 'bash tools/blas_solutions.sh r04'   ->  gpurun_out/prof/<tag>_blas_solutions.md
set -eo pipefail
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/blas" -o blas -- python3 "$ROOT/tools/blas_calib.py" > "$OUT/blas_calib.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/blas/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total Duration (ns)", 0)) or 0))
with open(f"{out}/{tag}_blas_solutions.md", "w") as w:
    w.write(f"# Vendor GEMM kernels behind torch.matmul on the step's hot shapes ({tag})\n\n")
    w.write("`rocprofv3 --kernel-trace --stats -- python tools/blas_calib.py` (each shape: 2 x 21 launches)\n\n")
    w.write("```\n" + open(out + "/blas_calib.log").read().strip() + "\n```\n\n| kernel | calls | avg us |\n|---|---|---|\n")
    for r in rows[:14]:
        name = r.get("Name") or r.get("KernelName")
        calls = r.get("Calls") or r.get("Count")
        avg = float(r.get("AverageNs") or r.get("Average (ns)") or 0) / 1e3
        w.write(f"| `{name}` | {calls} | {avg:.1f} |\n")
print(open(f"{out}/{tag}_blas_solutions.md").read())
PY
rm -rf "$OUT/blas"
