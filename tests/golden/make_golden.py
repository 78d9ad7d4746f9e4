#!/usr/bin/env python3
"""Regenerates tests/golden/synthetic_armours.json from the CPU oracle (oracle/) on the frozen synthetic
stream.  These are THE BUILD'S OWN regression vectors: the reference publishes none
and cannot be built here (OpenCV absent).  Floats are stored as C99 hex so the comparison is bit-exact."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from rmcv_amd import synth  # noqa: E402

O.set_math_mode(0)
W, H, CAMP = 1280, 1024, 1
out = {"width": W, "height": H, "camp": CAMP, "params": "executable/main.cpp:172-176 defaults", "frames": []}
for index, variant in [(0, 0), (1, 0), (2, 0), (3, 0), (1000, 1), (1001, 1), (1002, 1), (4242, 0)]:
    f = synth.frame(index, W, H, CAMP, variant)
    r = O.detect_frame(f)
    out["frames"].append({
        "index": index, "variant": variant, "frame_fnv1a": "%016x" % synth.checksum(f),
        "n_contours": len(r["offs"]) - 1, "n_points": len(r["pts"]), "n_blobs": len(r["blobs"]),
        "armour_vertices_hex": [[float.hex(float(v)) for v in a["vertices"].reshape(-1)] for a in r["armours"]],
    })
json.dump(out, open(os.path.join(HERE, "synthetic_armours.json"), "w"), indent=1)
print("wrote", len(out["frames"]), "frames")
