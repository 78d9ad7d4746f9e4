"""include/rmcv_shim.hpp -- the reference's rm:: signatures over the C-ABI -- compiled as C++17, LINKED the way
INTEGRATION.md prescribes (backend TU + unchanged caller TU, tests/shim/) and run end to end.
OpenCV is absent from this image, so the units are compiled against tests/cv_mock (a test-only stand-in for the few
cv:: types they touch; no reference source is built)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")
LIBDIR = os.path.join(ROOT, "rmcv_amd", "lib")


SHIM = os.path.join(HERE, "shim")
RM_SYMBOLS = ["rm::extract_color(", "rm::filter_lightblobs(", "rm::filter_armours(", "rm::MatchLightBlob(", "rm::FindLightBlobs(",
              "rm::LightBlobOverlap(", "rm::solve_PnP("]


def compile_units(tmp):
    """the layout INTEGRATION.md section 2 prescribes: backend TU (declarations + shim), the reference's own core TU
    (constructors; a stub here), and a caller TU that sees declarations only -- three separate objects"""
    objs = {}
    for unit in ("backend", "core_stub", "caller"):
        objs[unit] = os.path.join(tmp, unit + ".o")
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(HERE, "cv_mock"), "-I", os.path.join(ROOT, "include"),
                        "-I", SHIM, "-c", os.path.join(SHIM, unit + ".cpp"), "-o", objs[unit]], check=True)
    return objs


def build(tmp):
    objs = compile_units(tmp)
    exe = os.path.join(tmp, "shim_main")
    cmd = ["g++", objs["caller"], objs["backend"], objs["core_stub"], "-o", exe, "-L", LIBDIR, "-lrmcv_hip",
           "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True)
    return exe


def nm(obj, flag):
    return subprocess.run(["nm", "-C", flag, obj], check=True, capture_output=True, text=True).stdout


def test_backend_unit_defines_the_rm_symbols(tmp_path):
    """round 2's shim was all `inline`: a backend TU that does not call the functions emitted nothing and
    executable/main.cpp would not have linked.  The backend object must DEFINE (T) all seven; the caller object must
    leave the same seven UNDEFINED (U) -- it never saw the shim."""
    objs = compile_units(str(tmp_path))
    defined = nm(objs["backend"], "--defined-only")
    undefined = nm(objs["caller"], "--undefined-only")
    for sym in RM_SYMBOLS:
        assert any(sym in l and " T " in l for l in defined.splitlines()), "backend.o does not define " + sym
        assert any(sym in l for l in undefined.splitlines()), "caller.o does not reference " + sym
    assert "rmcv_extract_color" not in undefined  # the caller reaches the C-ABI only through rm::


def test_shim_links_as_two_translation_units(tmp_path):
    assert os.path.exists(build(str(tmp_path)))


@pytest.mark.gpu
def test_shim_matches_oracle(tmp_path, oracle):
    from rmcv_amd import synth
    exe = build(str(tmp_path))
    for index in (0, 5):
        out = subprocess.run([exe, str(index)], check=True, capture_output=True, text=True, timeout=120).stdout.strip().splitlines()
        ref = oracle.detect_frame(synth.frame(index))
        head = dict(zip(out[0].split()[0::2], map(int, out[0].split()[1::2])))
        assert head["contours"] == len(ref["offs"]) - 1 and head["points"] == len(ref["pts"])
        assert head["binary_on"] == int(np.count_nonzero(ref["binary"])) and head["positive"] == len(ref["blobs"])
        assert head["armours"] == len(ref["armours"])
        arm_lines = [l for l in out[1:] if l.startswith("armour")]
        got = [[float.fromhex(t) for t in line.split()[1:]] for line in arm_lines]
        exp = [[float(v) for v in a["vertices"].reshape(-1)] for a in ref["armours"]]
        assert got == exp
        poses = [[float.fromhex(t) for t in l.split()[1:]] for l in out if l.startswith("pose")]
        wr, wt, _ = oracle.locate_armours(ref["armours"])
        assert poses == [list(r) + list(t) for r, t in zip(wr.tolist(), wt.tolist())]
        # legacy matcher through the shim: FindLightBlobs(fitEllipse=false), MatchLightBlob(fitEllipse=true), LightBlobOverlap
        frame = synth.frame(index)
        lb, _, _ = oracle.find_lightblobs(frame, ref["pts"], ref["offs"], 1.5, 80, 70, 10, 99999, False)
        leg = [l for l in out if l.startswith("legacy")][0].split()
        assert int(leg[1]) == len(lb)
        vals = leg[2:]
        for i, b in enumerate(lb):
            assert int(vals[3 * i]) == int(b["target"])
            assert float.fromhex(vals[3 * i + 1]) == float(b["size"][0]) and float.fromhex(vals[3 * i + 2]) == float(b["size"][1])
        tail = [l for l in out if l.startswith("matched")][0].split()
        conts = [ref["pts"][ref["offs"][i]:ref["offs"][i + 1]] for i in range(len(ref["offs"]) - 1)]
        assert int(tail[1]) == sum(oracle.match_lightblob(c, 1.5, 80, 70, 10, 99999, True)[0] for c in conts)
        assert int(tail[3]) == sum(oracle.lightblob_overlap(lb, i, i + 2) == 1 for i in range(len(lb) - 2))
