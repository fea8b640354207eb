#!/usr/bin/env python3
"""Config 5: tiled VAE decode (257 frames 1280x736, z-tiling 4 + 1 latent frames) -- timing only, for rocprofv3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402

print(bench.time_vae("cuda", int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2, grid=(33, 23, 40), z_tile=4,
                     cross_check=sys.argv[-1] != "--no-check"), flush=True)
