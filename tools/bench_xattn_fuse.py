#!/usr/bin/env python3
"""bench.py --no-extras twice in one process: cross-attention's q normalised by a pass of its own / on load from a per-row
factor (ltxmi.attention.FUSE_CROSS_ATTENTION_Q).  Prints the two step times."""
import json
import os
import sys
import io
import contextlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ltx-video-gpupoor_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import ltxmi.attention as la  # noqa: E402

for fuse in (False, True, False, True):
    la.FUSE_CROSS_ATTENTION_Q = fuse
    sys.argv = ["bench.py", "--no-extras", "--steps", "20", "--warmup", "5"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"FUSE_CROSS_ATTENTION_Q={fuse}: {d['ms_per_step']} ms/step, {d['value']} steps/s", flush=True)
