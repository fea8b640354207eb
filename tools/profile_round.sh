#!/bin/bash
# The round's profile passes on the GPU box: kernel stats of the bench command, then the PMC passes (each in its own
# run, --pmc never combined with a trace domain).  Everything lands in gpurun_out/prof/; copy the summaries to profiles/.
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'
set -eo pipefail
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --steps 2 --warmup 1 --no-extras"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
python3 "$OLDPWD/tools/summarize_rocprof.py" "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.md" \
    "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-extras ($TAG)"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
echo "write done"
python3 "$OLDPWD/tools/pmc_traffic.py" "$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)" \
    "$(find "$OUT/write" -name '*counter_collection.csv' | head -1)" "$OUT/traffic.json" "$OUT/${TAG}_pmc_traffic.md" "profiles/${TAG}_pmc_traffic.md" "python bench.py --steps 2 --warmup 1 --no-extras"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/mfma" -o mfma -- $CMD > "$OUT/mfma.log" 2>&1
python3 "$OLDPWD/tools/pmc_mfma.py" "$(find "$OUT/mfma" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_mfma.md"
echo "mfma done"
# the raw CSVs are large: keep the summaries only
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/mfma"
ls -la "$OUT"
