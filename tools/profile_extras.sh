#!/bin/bash
# Side profiles of the round: the N = 98304 attention stress launch (MFMA busy, PMC) and a VAE decode (kernel stats).
#   gpurun --timeout 900 -- 'bash tools/profile_extras.sh r02'
set -eo pipefail
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/attn" -o attn -- python3 "$ROOT/tools/attn_once.py" 98304 > "$OUT/attn.log" 2>&1
python3 "$ROOT/tools/pmc_mfma.py" "$(find "$OUT/attn" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_mfma_attn98304.md"
echo "attention pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/vae" -o vae -- python3 "$ROOT/tools/vae_time.py" 5 --no-check > "$OUT/vae.log" 2>&1
python3 "$ROOT/tools/summarize_rocprof.py" "$(find "$OUT/vae" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_vae_kernel_stats.md" \
    "rocprofv3 --kernel-trace --stats -- python tools/vae_time.py 5 --no-check (VAE decode 768x512x97, $TAG)"
echo "vae stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/vaepmc" -o vaepmc -- python3 "$ROOT/tools/vae_time.py" 2 --no-check > "$OUT/vaepmc.log" 2>&1
python3 "$ROOT/tools/pmc_mfma.py" "$(find "$OUT/vaepmc" -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_pmc_vae.md"
# HBM-side traffic of a VAE decode (the convolution against BOTH roofs): FETCH_SIZE / WRITE_SIZE in passes of their own
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/vaefetch" -o vaefetch -- python3 "$ROOT/tools/vae_time.py" 2 --no-check > "$OUT/vaefetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/vaewrite" -o vaewrite -- python3 "$ROOT/tools/vae_time.py" 2 --no-check > "$OUT/vaewrite.log" 2>&1
python3 "$ROOT/tools/pmc_traffic.py" "$(find "$OUT/vaefetch" -name '*counter_collection.csv' | head -1)" \
    "$(find "$OUT/vaewrite" -name '*counter_collection.csv' | head -1)" "$OUT/traffic_vae.json" "$OUT/${TAG}_pmc_traffic_vae.md" "profiles/${TAG}_pmc_traffic_vae.md" "python tools/vae_time.py 2 (VAE decode 768x512x97)"
echo "vae traffic done"
# the same two passes over the z-tiled decode of config 5 (1280x720x257, tiles of 4 + 1 latent frames)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/vae5fetch" -o vae5fetch -- python3 "$ROOT/tools/vae_tiled_time.py" 1 --no-check > "$OUT/vae5fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/vae5write" -o vae5write -- python3 "$ROOT/tools/vae_tiled_time.py" 1 --no-check > "$OUT/vae5write.log" 2>&1
python3 "$ROOT/tools/pmc_traffic.py" "$(find "$OUT/vae5fetch" -name '*counter_collection.csv' | head -1)" \
    "$(find "$OUT/vae5write" -name '*counter_collection.csv' | head -1)" "$OUT/traffic_vae_config5.json" "$OUT/${TAG}_pmc_traffic_vae_config5.md" "profiles/${TAG}_pmc_traffic_vae_config5.md" "python tools/vae_tiled_time.py 1 (VAE decode 1280x720x257, z-tiled)"
echo "vae config-5 traffic done"
rm -rf "$OUT/attn" "$OUT/vae" "$OUT/vaepmc" "$OUT/vaefetch" "$OUT/vaewrite" "$OUT/vae5fetch" "$OUT/vae5write"
ls -la "$OUT"
