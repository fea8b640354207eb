#!/usr/bin/env python3
"""VAE decode timing only (bench.py's time_vae), for A/B runs with LTXMI_LIB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402

print(bench.time_vae("cuda", int(sys.argv[1]) if len(sys.argv) > 1 else 5), flush=True)
