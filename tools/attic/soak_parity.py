"""dev tool: the full-size parity check of tests/test_gpu_parity.py::test_full_size_every_stage_of_every_frame on MORE of the synthetic
stream than the suite has time for: `blocks` batches of 256 frames (1280x1024 and, every fourth, 1920x1200), both variants, both
sparse-kernel settings, all three camps that make sense on the stream -- every stage of every frame against the oracle.
usage: python tools/soak_parity.py [blocks] [first_seed]"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
from rmcv_amd import CAMP_BLUE, CAMP_RED, OPT_SPARSE_WAVES, Context, default_params, synth

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 900000
O.set_math_mode(0)
tot_f = tot_b = tot_a = 0
for b in range(blocks):
    w, h = (1920, 1200) if b % 4 == 3 else (1280, 1024)
    n = 128 if w == 1920 else 256
    variant, waves, camp = b % 2, (4 if b % 3 else 8), (CAMP_RED if b % 5 == 4 else CAMP_BLUE)
    frames = synth.batch(seed0 + 1000 * b, n, w, h, camp, variant, threads=16)
    c = Context(device=0, max_frames=n, max_width=w, max_height=h)
    c.set_option(OPT_SPARSE_WAVES, waves)
    gp = default_params(camp=camp)
    arm, offs = c.detect_batch(frames, gp)
    assert not (c.counts()["status"] & 15).any()
    p = O.default_params(); p.camp = camp
    with ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda f: O.detect_frame(frames[f], p), range(n)))
    for f in range(n):
        ref = refs[f]
        ok = np.array_equal(c.binary(f), ref["binary"])
        pts, co = c.contours(f)
        ok = ok and np.array_equal(co, ref["offs"]) and np.array_equal(pts, ref["pts"])
        blobs, _ = c.blobs(f)
        ok = ok and blobs.tobytes() == ref["blobs"].tobytes() and arm[offs[f]:offs[f + 1]].tobytes() == ref["armours"].tobytes()
        if not ok:
            np.save("gpurun_out/soak_parity_fail_%d_%d.npy" % (b, f), frames[f])
            print("MISMATCH block", b, "frame", f, "seed", seed0 + 1000 * b, w, h, "variant", variant, "waves", waves, "camp", camp)
            sys.exit(1)
        tot_b += len(blobs)
    tot_f += n; tot_a += int(offs[-1])
    c.close()
    print("block %d ok: %dx%d variant %d waves %d camp %d  (%d frames, %d blobs, %d armours so far)" % (b, w, h, variant, waves, camp, tot_f, tot_b, tot_a), flush=True)
print("soak ok:", tot_f, "frames,", tot_b, "light blobs,", tot_a, "armours, every stage bit-identical to the oracle")
