#!/usr/bin/env python3
"""Regenerates tests/golden/synthetic_armours.json from the CPU oracle (oracle/) on the frozen synthetic
stream.  These are THE BUILD'S OWN regression vectors: the reference publishes none
and cannot be built here (OpenCV absent).  Floats are stored as C99 hex so the comparison is bit-exact."""
import json
import os
import sys

import numpy as np

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

# the "next" rows of SURVEY 8f on the same frozen frames: icon identities (seeded stand-in SVM weights), legacy light blobs
# (FindLightBlobs with minAreaRect boxes and the camp vote), poses (solve_PnP + world position, default camera constants)
svm = synth.svm_weights()
nxt = {"width": W, "height": H, "camp": CAMP, "frames": []}
for index, variant in [(0, 0), (3, 0), (1000, 1), (4242, 0)]:
    f = synth.frame(index, W, H, CAMP, variant)
    r = O.detect_frame(f)
    ident, _, icons = O.classify_armours(f, r["armours"], svm)
    lb, src, boxes = O.find_lightblobs(f, r["pts"], r["offs"], 1.5, 80, 70, 10, 99999, False)
    rv, tv, pos = O.locate_armours(r["armours"])
    nxt["frames"].append({
        "index": index, "variant": variant,
        "identities": [int(v) for v in ident], "icon_sums": [int(ic.astype(np.int64).sum()) for ic in icons],
        "legacy_src": [int(v) for v in src], "legacy_targets": [int(b["target"]) for b in lb],
        "legacy_boxes_hex": [[float.hex(float(b[k])) for k in ("cx", "cy", "w", "h", "angle")] for b in boxes],
        "tvec_hex": [[float.hex(float(v)) for v in t] for t in tv], "position_hex": [[float.hex(float(v)) for v in q] for q in pos],
    })
json.dump(nxt, open(os.path.join(HERE, "synthetic_next_rows.json"), "w"), indent=1)
print("wrote next-row vectors for", len(nxt["frames"]), "frames")
