/*
 * synth.c -- deterministic, integer-only synthetic camera frames.
 *
 * The reference ships no data (the Daheng camera, hardware/src/daheng.cpp, is the
 * only source of frames), so the benchmark and the parity tests run on this
 * generator.  Everything is integer arithmetic (splitmix64 + a Q16 sin/cos table
 * for whole degrees), so any host produces bit-identical frames.
 *
 * Frame f of a stream: seed = 20241008 + f.
 *   background  every byte uniform in [0,48)                -> |A-B| <= 47 never reaches lb=80
 *   armours     K in 1..4: two parallel bars, height h in 24..96, width max(3,h/6),
 *               centre gap in [1.2h,3h], common tilt -10..10 deg, per-bar jitter -3..3 deg,
 *               >= 8 px from the frame edge, not overlapping; enemy colour
 *               (blue: B=255, G in 160..220, R in 0..60; red: mirrored)
 *   distractors D in 0..6 bars of the friendly colour (never fire), tilt -30..30
 *   specks      S in 0..20 enemy-colour specks of 1..3 px (area < 10: exercise the gate)
 *   variant 1 ("stress"): a white core of width/3 inside each enemy bar (holes / splits)
 *               and 0.1 % white salt pixels.
 *   variants 10..14 ("dense", level L = variant - 10): the plain stream plus what a real camera adds to it -- bright
 *               enemy-coloured areas (lit windows, reflections on the floor) and far more small specks (glare on edges, LEDs):
 *               level     :    0     1     2     3     4
 *               specks    :   +0  +100  +400 +1000 +2000   (1..3 px, on top of the plain stream's 0..20; exact counts)
 *               windows   :    0     1     3     7    13   rectangles of 40..120 x 30..100 px with a ragged (noisy) rim
 *               foreground:  0.25   0.6   1.4   2.9   5.3  % of the pixels at 1280x1024 (level 0 = the plain stream)
 *               Windows and specks may touch bars and each other (nothing is kept apart): a frame of level 3 has ~950
 *               contours, ~6000 border points and ~3000 non-empty words -- beyond what findContours' LDS tables
 *               hold, which is what these variants are for (cv::findContours itself has no bound, src/imgproc.cpp:71-72).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int32_t k_sin_q16[46] = {0, 1144, 2287, 3430, 4572, 5712, 6850, 7987, 9121, 10252, 11380, 12505,
    13626, 14742, 15855, 16962, 18064, 19161, 20252, 21336, 22415, 23486, 24550, 25607, 26656, 27697, 28729,
    29753, 30767, 31772, 32768, 33754, 34729, 35693, 36647, 37590, 38521, 39441, 40348, 41243, 42126, 42995,
    43852, 44695, 45525, 46341};
static const int32_t k_cos_q16[46] = {65536, 65526, 65496, 65446, 65376, 65287, 65177, 65048, 64898, 64729,
    64540, 64332, 64104, 63856, 63589, 63303, 62997, 62672, 62328, 61966, 61584, 61183, 60764, 60326, 59870,
    59396, 58903, 58393, 57865, 57319, 56756, 56175, 55578, 54963, 54332, 53684, 53020, 52339, 51643, 50931,
    50203, 49461, 48703, 47930, 47143, 46341};

typedef struct { uint64_t s; } rng_t;
static uint64_t rng_next(rng_t* r)
{ /* splitmix64 */
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int rng_range(rng_t* r, int lo, int hi) /* inclusive */
{
    return lo + (int)(rng_next(r) % (uint64_t)(hi - lo + 1));
}

typedef struct { int x0, y0, x1, y1; } box_t;

/* paint a rotated bar: centre (cx,cy), full height bh (along the bar), full width bw,
 * angle a degrees from vertical (positive = top leans right).  core_w > 0 paints a white
 * core of that width.  Returns the bounding box. */
static box_t paint_bar(uint8_t* img, int w, int h, int stride, int cx, int cy, int bh, int bw, int a,
                       uint8_t b, uint8_t g, uint8_t r, int core_w, int do_paint)
{
    int aa = a < 0 ? -a : a;
    int32_t sn = k_sin_q16[aa], cs = k_cos_q16[aa];
    if (a < 0) sn = -sn;
    /* half extents in Q16, measured in half-pixels so odd sizes stay centred */
    int64_t hl = (int64_t)bh << 15, hw = (int64_t)bw << 15, hc = (int64_t)core_w << 15;
    int rad = (bh + bw) / 2 + 2;
    box_t bb = {cx - rad, cy - rad, cx + rad, cy + rad};
    box_t used = {w, h, -1, -1};
    for (int y = bb.y0; y <= bb.y1; y++) {
        if (y < 0 || y >= h) continue;
        for (int x = bb.x0; x <= bb.x1; x++) {
            if (x < 0 || x >= w) continue;
            int64_t dx = x - cx, dy = y - cy;
            /* u across the bar, v along it; bar axis = (sin a, -cos a) in image coordinates */
            int64_t u = dx * cs + dy * sn;
            int64_t v = -dx * sn + dy * cs;
            if (u < 0) u = -u;
            if (v < 0) v = -v;
            if (u <= hw && v <= hl) {
                if (do_paint) {
                    uint8_t* p = img + (size_t)y * stride + 3 * (size_t)x;
                    if (core_w > 0 && u <= hc && v <= hl - (2 << 16)) { p[0] = p[1] = p[2] = 255; }
                    else { p[0] = b; p[1] = g; p[2] = r; }
                }
                if (x < used.x0) used.x0 = x;
                if (x > used.x1) used.x1 = x;
                if (y < used.y0) used.y0 = y;
                if (y > used.y1) used.y1 = y;
            }
        }
    }
    return used;
}

static int boxes_overlap(const box_t* a, const box_t* b, int margin)
{
    return !(a->x1 + margin < b->x0 || b->x1 + margin < a->x0 || a->y1 + margin < b->y0 || b->y1 + margin < a->y0);
}

/* camp: 1 = blue enemy, otherwise red enemy (the mirrored stream). variant: 0 plain, 1 stress. */
int rmcv_synth_frame(uint8_t* bgr, int w, int h, int stride, uint64_t frame_index, int camp, int variant)
{
    if (!bgr || w < 64 || h < 64 || stride < 3 * w) return -1;
    const uint64_t seed = 20241008ull + frame_index;
    rng_t bg = {seed * 0x9E3779B97F4A7C15ull + 1};
    rng_t ob = {seed * 0xD1B54A32D192ED03ull + 2};

    /* background */
    for (int y = 0; y < h; y++) {
        uint8_t* row = bgr + (size_t)y * stride;
        int n = 3 * w, i = 0;
        while (i < n) {
            uint64_t v = rng_next(&bg);
            for (int k = 0; k < 8 && i < n; k++, i++) row[i] = (uint8_t)((((v >> (8 * k)) & 0xFF) * 48) >> 8);
        }
    }

    box_t placed[64];
    int n_placed = 0;

    /* armours */
    int K = rng_range(&ob, 1, 4);
    for (int k = 0; k < K; k++) {
        int bh = rng_range(&ob, 24, 96);
        int bw = bh / 6 < 3 ? 3 : bh / 6;
        int gap = rng_range(&ob, bh * 12 / 10, bh * 3);
        int tilt = rng_range(&ob, -10, 10);
        int j0 = rng_range(&ob, -3, 3), j1 = rng_range(&ob, -3, 3);
        uint8_t g0 = (uint8_t)rng_range(&ob, 160, 220), r0 = (uint8_t)rng_range(&ob, 0, 60);
        uint8_t g1 = (uint8_t)rng_range(&ob, 160, 220), r1 = (uint8_t)rng_range(&ob, 0, 60);
        int ex = gap / 2 + bh / 2 + bw + 10, ey = bh / 2 + bw + gap / 4 + 10;
        int ok = 0, cx = 0, cy = 0;
        box_t b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        for (int attempt = 0; attempt < 40 && !ok; attempt++) {
            if (w - 2 * ex <= 0 || h - 2 * ey <= 0) { (void)rng_next(&ob); (void)rng_next(&ob); continue; }
            cx = rng_range(&ob, ex, w - 1 - ex);
            cy = rng_range(&ob, ey, h - 1 - ey);
            /* the pair is laid out along the direction perpendicular to the common tilt */
            int at = tilt < 0 ? -tilt : tilt;
            int32_t sn = tilt < 0 ? -k_sin_q16[at] : k_sin_q16[at], cs = k_cos_q16[at];
            int ox = (int)(((int64_t)(gap / 2) * cs) >> 16), oy = (int)(((int64_t)(gap / 2) * sn) >> 16);
            b0 = paint_bar(bgr, w, h, stride, cx - ox, cy - oy, bh, bw, tilt + j0, 0, 0, 0, 0, 0);
            b1 = paint_bar(bgr, w, h, stride, cx + ox, cy + oy, bh, bw, tilt + j1, 0, 0, 0, 0, 0);
            ok = b0.x0 >= 8 && b0.y0 >= 8 && b0.x1 < w - 8 && b0.y1 < h - 8 && b1.x0 >= 8 && b1.y0 >= 8 &&
                 b1.x1 < w - 8 && b1.y1 < h - 8 && !boxes_overlap(&b0, &b1, 3);
            for (int q = 0; q < n_placed && ok; q++)
                if (boxes_overlap(&b0, &placed[q], 4) || boxes_overlap(&b1, &placed[q], 4)) ok = 0;
            if (ok && n_placed + 2 <= 64) {
                int core = variant == 1 ? (bw / 3 < 1 ? 1 : bw / 3) : 0;
                paint_bar(bgr, w, h, stride, cx - ox, cy - oy, bh, bw, tilt + j0, 255, g0, r0, core, 1);
                paint_bar(bgr, w, h, stride, cx + ox, cy + oy, bh, bw, tilt + j1, 255, g1, r1, core, 1);
                placed[n_placed++] = b0;
                placed[n_placed++] = b1;
            } else ok = 0;
        }
    }

    /* friendly-colour distractor bars */
    int D = rng_range(&ob, 0, 6);
    for (int d = 0; d < D; d++) {
        int bh = rng_range(&ob, 24, 96);
        int bw = bh / 6 < 3 ? 3 : bh / 6;
        int tilt = rng_range(&ob, -30, 30);
        uint8_t g0 = (uint8_t)rng_range(&ob, 160, 220), b0c = (uint8_t)rng_range(&ob, 0, 60);
        int ex = bh / 2 + bw + 10;
        for (int attempt = 0; attempt < 20; attempt++) {
            if (w - 2 * ex <= 0 || h - 2 * ex <= 0) { (void)rng_next(&ob); (void)rng_next(&ob); continue; }
            int cx = rng_range(&ob, ex, w - 1 - ex), cy = rng_range(&ob, ex, h - 1 - ex);
            box_t b = paint_bar(bgr, w, h, stride, cx, cy, bh, bw, tilt, 0, 0, 0, 0, 0);
            int ok = 1;
            for (int q = 0; q < n_placed && ok; q++)
                if (boxes_overlap(&b, &placed[q], 4)) ok = 0;
            if (ok && n_placed < 64) {
                paint_bar(bgr, w, h, stride, cx, cy, bh, bw, tilt, b0c, g0, 255, 0, 1);
                placed[n_placed++] = b;
                break;
            }
        }
    }

    /* enemy-colour specks of 1..3 px */
    int S = rng_range(&ob, 0, 20);
    for (int s = 0; s < S; s++) {
        int x = rng_range(&ob, 1, w - 3), y = rng_range(&ob, 1, h - 3), npx = rng_range(&ob, 1, 3);
        uint8_t g0 = (uint8_t)rng_range(&ob, 160, 220), r0 = (uint8_t)rng_range(&ob, 0, 60);
        static const int sx[3] = {0, 1, 0}, sy[3] = {0, 0, 1};
        for (int k = 0; k < npx; k++) {
            uint8_t* p = bgr + (size_t)(y + sy[k]) * stride + 3 * (size_t)(x + sx[k]);
            p[0] = 255; p[1] = g0; p[2] = r0;
        }
    }

    if (variant >= 10 && variant <= 14) { /* dense levels */
        static const int n_specks[5] = {0, 100, 400, 1000, 2000}, n_windows[5] = {0, 1, 3, 7, 13};
        const int L = variant - 10;
        rng_t dn = {seed * 0xA24BAED4963EE407ull + 3};
        for (int k = 0; k < n_windows[L]; k++) {
            int ww_ = rng_range(&dn, 40, 120), wh_ = rng_range(&dn, 30, 100);
            if (ww_ > w - 4) ww_ = w - 4;
            if (wh_ > h - 4) wh_ = h - 4;
            int x0 = rng_range(&dn, 1, w - 2 - ww_), y0 = rng_range(&dn, 1, h - 2 - wh_);
            uint8_t g0 = (uint8_t)rng_range(&dn, 160, 220), r0 = (uint8_t)rng_range(&dn, 0, 60);
            for (int y = y0; y < y0 + wh_; y++)
                for (int x = x0; x < x0 + ww_; x++) {
                    /* ragged rim: the outermost two pixel rings are lit with probability 1/2 */
                    int rim = (y - y0 < 2) || (y0 + wh_ - 1 - y < 2) || (x - x0 < 2) || (x0 + ww_ - 1 - x < 2);
                    if (rim && (rng_next(&dn) & 1)) continue;
                    uint8_t* p = bgr + (size_t)y * stride + 3 * (size_t)x;
                    p[0] = 255; p[1] = g0; p[2] = r0;
                }
        }
        for (int s2 = 0; s2 < n_specks[L]; s2++) {
            int x = rng_range(&dn, 1, w - 3), y = rng_range(&dn, 1, h - 3), npx = rng_range(&dn, 1, 3);
            uint8_t g0 = (uint8_t)rng_range(&dn, 160, 220), r0 = (uint8_t)rng_range(&dn, 0, 60);
            static const int sx[3] = {0, 1, 0}, sy[3] = {0, 0, 1};
            for (int k = 0; k < npx; k++) {
                uint8_t* p = bgr + (size_t)(y + sy[k]) * stride + 3 * (size_t)(x + sx[k]);
                p[0] = 255; p[1] = g0; p[2] = r0;
            }
        }
    }

    if (variant == 1) { /* salt */
        int n = (int)(((int64_t)w * h) / 1000);
        for (int i = 0; i < n; i++) {
            int x = rng_range(&ob, 0, w - 1), y = rng_range(&ob, 0, h - 1);
            uint8_t* p = bgr + (size_t)y * stride + 3 * (size_t)x;
            p[0] = p[1] = p[2] = 255;
        }
    }

    if (camp != 1) { /* mirrored stream: swap B and R */
        for (int y = 0; y < h; y++) {
            uint8_t* row = bgr + (size_t)y * stride;
            for (int x = 0; x < w; x++) {
                uint8_t t = row[3 * x];
                row[3 * x] = row[3 * x + 2];
                row[3 * x + 2] = t;
            }
        }
    }
    return 0;
}

/* FNV-1a over the frame: the committed checksums in tests/golden/ pin the generator. */
uint64_t rmcv_synth_checksum(const uint8_t* bgr, int w, int h, int stride)
{
    uint64_t hsh = 0xcbf29ce484222325ull;
    for (int y = 0; y < h; y++) {
        const uint8_t* row = bgr + (size_t)y * stride;
        for (int i = 0; i < 3 * w; i++) { hsh ^= row[i]; hsh *= 0x100000001b3ull; }
    }
    return hsh;
}
