#!/bin/bash
# HBM-side traffic of ONE self-attention launch shape under two library builds (FETCH_SIZE / WRITE_SIZE in passes of their own).
#   gpurun -- 'bash tools/attn_traffic_ab.sh libA.so libB.so [N [B]]'
set -eo pipefail
ROOT=$PWD; OUT=$PWD/gpurun_out/attn_traffic; mkdir -p "$OUT"; export TMPDIR=/tmp
N=${3:-4992}; B=${4:-3}
cd /tmp
for lib in "$1" "$2"; do
  tag=$(basename "$lib" .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    LTXMI_LIB="$ROOT/$lib" rocprofv3 --pmc $c --output-format csv -d "$OUT/${tag}_$c" -o x -- python3 "$ROOT/tools/attn_once.py" $N $B > "$OUT/${tag}_$c.log" 2>&1
  done
  python3 - "$OUT" "$tag" <<'PY'
import csv, sys, glob, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
for c, mul in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):      # KiB; FETCH counts 64 B per 128-B request on gfx950
    f = glob.glob(f"{out}/{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "attn" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[c] = mul * sum(v) / len(v)
for k, v in res.items():
    print(f"{tag}: {k}: fetch {v.get('FETCH_SIZE', 0) / 1e6:.1f} MB + write {v.get('WRITE_SIZE', 0) / 1e6:.1f} MB per launch")
PY
done
rm -rf "$OUT"/*_FETCH_SIZE "$OUT"/*_WRITE_SIZE
