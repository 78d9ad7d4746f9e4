"""BASELINE config 5 / SURVEY 8f-1: icon rectification (rm::affine_correction) + flatten + linear SVM vote.
CPU part: hand-derived KATs of the oracle's restatement.  GPU part: bit-exact parity of icons and identities."""
import numpy as np
import pytest

import oracle_lib as O


def test_svm_vote_kat():
    """one-vs-one vote with hand-made weights: class pair (i<j) votes i when w.x - rho > 0"""
    n_class, nf = 3, 1200
    w = np.zeros((3, nf), np.float32)
    rho = np.zeros(3, np.float64)
    labels = np.array([10, 20, 30], np.int32)
    x = np.zeros(nf, np.float32)
    x[0], x[1] = 2.0, 5.0

    def predict():
        return O.lib().orc_svm_predict(O._p(x), nf, O._p(w), O._p(rho), O._p(labels), n_class)
    # all sums are 0 -> "sum > 0" false -> every pair votes j: votes = [0, 1, 2] -> class 2
    assert predict() == 30
    w[0, 0] = 1.0          # (0,1): 2 > 0 -> vote 0; (0,2) -> 2; (1,2) -> 2  => [1,0,2] -> 30
    assert predict() == 30
    w[1, 1] = 1.0          # (0,2): 5 > 0 -> vote 0                          => [2,0,1] -> 10
    assert predict() == 10
    rho[1] = 5.0           # (0,2): 5 - 5 = 0 -> not > 0 -> vote 2           => [1,0,2] -> 30
    assert predict() == 30
    rho[:] = 0
    w[:] = 0
    w[2, 0] = 1.0          # (1,2) -> vote 1 ; ties: first maximum wins: votes [0,2,1] -> 20
    assert predict() == 20


def test_affine_correction_kats():
    img = np.full((200, 300, 3), 100, np.uint8)
    img[..., 1] = 50
    img[..., 2] = 200
    # axis-aligned icon well inside the frame: every output pixel whose taps stay inside the ROI is the constant colour
    icon = np.array([[50, 120], [50, 40], [130, 40], [130, 120]], np.float32)   # [0]=bottom-left [1]=top-left [2]=top-right
    out, ic, rc = O.affine_correction(img, icon)
    assert rc == 0 and np.array_equal(ic, icon)
    assert (out[1:-1, 1:-1] == np.array([100, 50, 200], np.uint8)).all()
    # clamping happens in place (imgproc.cpp:11-15)
    icon2 = np.array([[-20, 250], [-20, 40], [330, 40], [330, 250]], np.float32)
    out2, ic2, rc = O.affine_correction(img, icon2)
    assert rc == 0 and ic2.tolist() == [[0, 199], [0, 40], [299, 40], [299, 199]]
    # exact 2:1 ROI (40x40 px) takes the INTER_AREA branch: a vertical edge image averages 2x2 boxes
    e = np.zeros((100, 100, 3), np.uint8)
    e[:, 30:] = 255
    icon3 = np.array([[10, 49], [10, 10], [49, 10], [49, 49]], np.float32)     # box x 10..49, y 10..49 -> 40x40
    out3, _, rc = O.affine_correction(e, icon3)
    assert rc == 0 and out3.shape == (20, 20, 3)
    assert out3[5, 2, 0] == 0 and out3[5, 15, 0] == 255
    # a tilted icon samples a gradient monotonically
    g = np.zeros((120, 160, 3), np.uint8)
    g[..., 0] = np.arange(160, dtype=np.uint8)[None, :]
    icon4 = np.array([[40, 90], [50, 30], [110, 40], [100, 100]], np.float32)
    out4, _, rc = O.affine_correction(g, icon4)
    row = out4[10, 2:-2, 0].astype(int)
    assert rc == 0 and (np.diff(row) >= 0).all() and row[-1] > row[0] + 30


def test_classify_is_deterministic_on_stream():
    from rmcv_amd import synth
    svm = synth.svm_weights()
    f = synth.frame(3, 1920, 1200)
    arm = O.detect_frame(f)["armours"]
    assert len(arm) > 0
    a, _, ia = O.classify_armours(f, arm, svm)
    b, _, ib = O.classify_armours(f, arm, svm)
    assert np.array_equal(a, b) and np.array_equal(ia, ib) and set(a.tolist()) <= set(range(7))


@pytest.mark.gpu
def test_classify_parity_single_frame(ctx, oracle):
    from rmcv_amd import synth
    svm = synth.svm_weights()
    ctx.svm_load(*svm)
    total = 0
    for idx, (w, h) in [(3, (1920, 1200)), (4, (1920, 1200)), (1005, (1280, 1024)), (9, (1280, 1024))]:
        f = synth.frame(idx, w, h, variant=1 if idx > 1000 else 0)
        arm = oracle.detect_frame(f)["armours"]
        ri, ra, ricons = oracle.classify_armours(f, arm, svm)
        gi, ga, gicons = ctx.classify_armours(f, arm)
        assert np.array_equal(gicons, ricons), idx
        assert np.array_equal(gi, ri), idx
        assert ga.tobytes() == ra.tobytes(), idx      # icon vertices clamped in place, like the reference
        total += len(arm)
    assert total > 5
    # icons that leave the frame are clamped; a degenerate (zero-area) icon gives a zero image, not a crash
    f = synth.frame(3, 640, 480)
    weird = np.zeros(3, oracle.ARMOUR)
    weird[0]["icon"] = [[-50, 600], [-50, -30], [700, -30], [700, 600]]
    weird[1]["icon"] = [[100, 100], [100, 100], [100, 100], [100, 100]]
    weird[2]["icon"] = [[300, 200], [310, 150], [380, 170], [370, 220]]
    ri, ra, ricons = oracle.classify_armours(f, weird, svm)
    gi, ga, gicons = ctx.classify_armours(f, weird)
    assert np.array_equal(gicons, ricons) and np.array_equal(gi, ri) and ga.tobytes() == ra.tobytes()


@pytest.mark.gpu
def test_classify_parity_batch_c5(oracle):
    """BASELINE config 5 at test size: 1920x1200 frames, full path + identity stage"""
    from rmcv_amd import STAGE_ALL, STAGE_IDENTITY, Context, default_params, synth
    svm = synth.svm_weights()
    n = 6
    big = Context(device=0, max_frames=n, max_width=1920, max_height=1200)
    big.svm_load(*svm)
    frames = synth.batch(40, n, 1920, 1200)
    big.upload(frames)
    big.run(default_params(), STAGE_ALL | STAGE_IDENTITY)
    big.sync()
    arm, offs = big.armours()
    ident = big.identities()
    assert len(ident) == len(arm) and len(arm) > n
    for f in range(n):
        ref = oracle.detect_frame(frames[f])
        ri, ra, ricons = oracle.classify_armours(frames[f], ref["armours"], svm)
        assert arm[offs[f]:offs[f + 1]].tobytes() == ra.tobytes(), f
        assert np.array_equal(ident[offs[f]:offs[f + 1]], ri), f
        assert np.array_equal(big.icons(f), ricons), f
    big.close()
