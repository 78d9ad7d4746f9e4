/*
 * rmcv_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See rmcv_oracle.h for scope and parity status ("parity unpinned" vs OpenCV).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  No FMA
 * contraction, strict IEEE double/float, sequential accumulation everywhere.
 *
 * Each function cites the reference line it restates.  OpenCV behaviour that
 * cannot be cited into /root/reference (OpenCV is not vendored) is marked
 * [OCV] and restates the published algorithm of OpenCV 4.8 (modules/imgproc:
 * morph, contours (Suzuki-Abe 1985), shapedescr (Fitzgibbon/Halir-Flusser
 * direct ellipse fit), modules/core: JAMA-derived eigenNonSymmetric).
 */
#include "rmcv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../rmcv_amd/csrc/pinned_math.h"

/* ------------------------------------------------------------------ math mode */
static int g_math_mode = 0;
void orc_set_math_mode(int mode) { g_math_mode = mode ? 1 : 0; }
int orc_get_math_mode(void) { return g_math_mode; }

static double m_sin(double x) { return g_math_mode ? sin(x) : pm_sin(x); }
static double m_cos(double x) { return g_math_mode ? cos(x) : pm_cos(x); }
static double m_atan2(double y, double x) { return g_math_mode ? atan2(y, x) : pm_atan2(y, x); }
static float m_atan2f(float y, float x) { return g_math_mode ? atan2f(y, x) : pm_atan2f(y, x); }
static float m_sinf(float x) { return g_math_mode ? sinf(x) : pm_sinf(x); }
static float m_cosf(float x) { return g_math_mode ? cosf(x) : pm_cosf(x); }
static double m_fmod180(double x) { return g_math_mode ? fmod(x, 180.0) : pm_fmod180(x); }

/* the reference calls abs / atan2 / sin / cos UNQUALIFIED on floats (src/objdetect.cpp:24,79,131-143,153,157,
 * src/core.cpp:335-337), and which function that is depends on the headers its translation unit happens to see.  Mode bits:
 *   bit 0  abs(float) resolves to int abs(int) (<cmath> alone under libstdc++: the argument is truncated towards zero)
 *   bit 1  atan2 / sin / cos on floats resolve to the double functions of <math.h> (the arithmetic around them is then double too)
 * 0 (default): the float overloads everywhere (libc++; libstdc++ once <stdlib.h> and <math.h> are in sight).
 * INTEGRATION.md shows the three-line probe that tells a maintainer which mode their build is. */
static int g_overloads = 0;
void orc_set_overload_mode(int mode) { g_overloads = mode & 3; }
int orc_get_overload_mode(void) { return g_overloads; }
float orc_abs_ov(float x) { return (g_overloads & 1) ? (float)abs((int)x) : fabsf(x); }
/* `atan2(y, x) * 180.0f / static_cast<float>(CV_PI)` assigned to a float (objdetect.cpp:137) */
static float ov_atan2_deg(float y, float x)
{
    const float pi_f = (float)3.1415926535897932384626433832795; /* static_cast<float>(CV_PI) */
    if (g_overloads & 2) return (float)(m_atan2((double)y, (double)x) * (double)180.0f / (double)pi_f);
    return m_atan2f(y, x) * 180.0f / pi_f;
}

#define ORC_PI 3.1415926535897932384626433832795 /* CV_PI */

void orc_default_params(orc_params* p)
{ /* executable/main.cpp:172-176 */
    memset(p, 0, sizeof(*p));
    p->camp = ORC_CAMP_BLUE;
    p->lower_bound = 80;
    p->morph = ORC_MORPH_CLOSE;
    p->tilt_max = 70.0f;
    p->ratio_lo = 1.5f;
    p->ratio_hi = 80.0f;
    p->area_lo = 10.0;
    p->area_hi = 99999.0;
    p->angle_diff_max = 12.0f;
    p->shear_max = 22.0f;
    p->length_ratio_max = 0.4f;
}

/* ------------------------------------------------- imgproc.cpp:52-69 (pixel part) */

/* [OCV] cv::dilate / cv::erode, 3x3 rect, anchor centre, BORDER_CONSTANT with the default morphology border
 * value: out-of-image samples never win (they are simply left out of the max / min).  A 3x3 box max (min) is
 * separable: rows first, then columns -- same result as the 9-tap form, three passes fewer. */
static void morph3x3(const uint8_t* in, uint8_t* out, int w, int h, int is_max)
{
    uint8_t* tmp = (uint8_t*)malloc((size_t)w * h);
    if (!tmp) return;
    for (int y = 0; y < h; y++) {
        const uint8_t* r = in + (size_t)y * w;
        uint8_t* t = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            uint8_t m = r[x];
            if (x > 0) m = is_max ? (r[x - 1] > m ? r[x - 1] : m) : (r[x - 1] < m ? r[x - 1] : m);
            if (x + 1 < w) m = is_max ? (r[x + 1] > m ? r[x + 1] : m) : (r[x + 1] < m ? r[x + 1] : m);
            t[x] = m;
        }
    }
    for (int y = 0; y < h; y++) {
        const uint8_t* t1 = tmp + (size_t)y * w;
        const uint8_t* t0 = y > 0 ? t1 - w : t1;
        const uint8_t* t2 = y + 1 < h ? t1 + w : t1;
        uint8_t* o = out + (size_t)y * w;
        if (is_max)
            for (int x = 0; x < w; x++) {
                uint8_t m = t0[x] > t1[x] ? t0[x] : t1[x];
                o[x] = t2[x] > m ? t2[x] : m;
            }
        else
            for (int x = 0; x < w; x++) {
                uint8_t m = t0[x] < t1[x] ? t0[x] : t1[x];
                o[x] = t2[x] < m ? t2[x] : m;
            }
    }
    free(tmp);
}
void orc_dilate3x3(const uint8_t* in, uint8_t* out, int w, int h) { morph3x3(in, out, w, h, 1); }
void orc_erode3x3(const uint8_t* in, uint8_t* out, int w, int h) { morph3x3(in, out, w, h, 0); }

int orc_extract_binary(const uint8_t* bgr, int w, int h, int stride, int camp, int lower_bound, int morph,
                       uint8_t* binary)
{
    if (!bgr || !binary || w <= 0 || h <= 0 || stride < 3 * w) return -1;
    /* imgproc.cpp:52-65: split; gray = chA - chB (cv::subtract on CV_8U saturates);
     * inRange(gray, lower_bound, 255) is inclusive on both ends. */
    int ca, cb; /* channel indices in BGR order */
    if (camp == ORC_CAMP_GUIDELIGHT) { ca = 1; cb = 2; }      /* G - R, :58 */
    else if (camp == ORC_CAMP_BLUE)  { ca = 0; cb = 2; }      /* B - R, :63 */
    else                             { ca = 2; cb = 0; }      /* R - B, :63 (RED, NEUTRAL, anything else) */
    for (int y = 0; y < h; y++) {
        const uint8_t* row = bgr + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            int d = (int)row[3 * x + ca] - (int)row[3 * x + cb];
            int gray = d < 0 ? 0 : d; /* saturate_cast<uchar> */
            binary[(size_t)y * w + x] = (gray >= lower_bound && gray <= 255) ? 255 : 0;
        }
    }
    if (morph == ORC_MORPH_NONE) return 0;
    /* imgproc.cpp:68-69: MORPH_CLOSE = dilate then erode with the same 3x3 kernel */
    uint8_t* tmp = (uint8_t*)malloc((size_t)w * h);
    if (!tmp) return -3;
    orc_dilate3x3(binary, tmp, w, h);
    if (morph == ORC_MORPH_DILATE) memcpy(binary, tmp, (size_t)w * h);
    else orc_erode3x3(tmp, binary, w, h);
    free(tmp);
    return 0;
}

/* ------------------------------------------------- imgproc.cpp:71-72 findContours */

/* [OCV] cv::findContours(binary, RETR_EXTERNAL, CHAIN_APPROX_NONE): the legacy
 * Suzuki-Abe raster scan (contours.cpp, cvFindNextContour + icvFetchContour).
 *  - image copied into a buffer with a 1-pixel zero frame, non-zero -> 1
 *  - outer border starts where prev==0 && p==1; in external mode it is kept
 *    only if the pixel at `lnbd` (last labelled pixel met on this row) is not
 *    positive; hole borders are never traced in this mode
 *  - tracing marks pixels 2, or (2|-128) when the neighbour sweep passed the
 *    east neighbour as zero; CHAIN_APPROX_NONE emits the point at every step
 *  - output order is the reverse of discovery order. */
static const int k_dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int k_dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

typedef struct {
    orc_point* p;
    int n, cap;
} ptvec;
static int pv_push(ptvec* v, int x, int y)
{
    if (v->n == v->cap) {
        int nc = v->cap ? v->cap * 2 : 1024;
        orc_point* np = (orc_point*)realloc(v->p, (size_t)nc * sizeof(orc_point));
        if (!np) return -1;
        v->p = np;
        v->cap = nc;
    }
    v->p[v->n].x = x;
    v->p[v->n].y = y;
    v->n++;
    return 0;
}

/* icvFetchContour, outer border (is_hole = 0), method CHAIN_APPROX_NONE.
 * img points at the padded image, step its row pitch, (px,py) padded coords. */
static int trace_border(signed char* img, int step, int px, int py, ptvec* out)
{
    const signed char nbd = 2;
    int deltas[16];
    for (int k = 0; k < 8; k++) deltas[k] = deltas[k + 8] = k_dy[k] * step + k_dx[k];
    signed char* i0 = img + (size_t)py * step + px;
    signed char *i1, *i3, *i4 = 0;
    int s, s_end;
    int x = px - 1, y = py - 1; /* emitted in original image coordinates (offset (-1,-1)) */

    s_end = s = 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);

    if (s == s_end) { /* single pixel domain */
        *i0 = (signed char)(nbd | -128);
        return pv_push(out, x, y);
    }
    i3 = i0;
    for (;;) {
        s_end = s;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        /* check "right" bound */
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
        else if (*i3 == 1) *i3 = nbd;
        if (pv_push(out, x, y)) return -1;
        x += k_dx[s];
        y += k_dy[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return 0;
}

int orc_find_contours(const uint8_t* binary, int w, int h, orc_point* pts, int cap_pts, int32_t* offs,
                      int cap_contours, int32_t* n_contours, int32_t* n_points)
{
    if (!binary || w <= 0 || h <= 0) return -1;
    const int step = w + 2, H = h + 2;
    signed char* img = (signed char*)calloc((size_t)step * H, 1);
    if (!img) return -3;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = binary[(size_t)y * w + x] ? 1 : 0;

    ptvec all = {0, 0, 0};
    int* starts = 0; /* discovery-order offsets */
    int n = 0, cap_n = 0, rc = 0;

    const int width = step - 1, height = H - 1;
    for (int y = 1; y < height; y++) {
        signed char* row = img + (size_t)y * step;
        int lnbd_x = 0; /* lnbd.y is always the current row */
        int prev = 0;
        int x = 1;
        for (; x < width; x++) {
            int p = 0;
            for (; x < width && (p = row[x]) == prev; x++) {}
            if (x >= width) break;
            int is_outer = (prev == 0 && p == 1);
            if (!is_outer) {
                /* hole check: (p != 0 || prev < 1) -> resume; a hole border is never
                 * traced in RETR_EXTERNAL (mode 0), so both branches resume the scan. */
                goto resume_scan;
            }
            if (row[lnbd_x] > 0) goto resume_scan; /* mode == 0 && img0[lnbd] > 0 */
            lnbd_x = x;
            if (n == cap_n) {
                int nc = cap_n ? cap_n * 2 : 256;
                int* ns = (int*)realloc(starts, (size_t)(nc + 1) * sizeof(int));
                if (!ns) { rc = -3; goto done; }
                starts = ns;
                cap_n = nc;
            }
            starts[n++] = all.n;
            if (trace_border(img, step, x, y, &all)) { rc = -3; goto done; }
            /* the scan resumes at x+1 with prev = img[x] (the label just written) */
            p = row[x];
        resume_scan:
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    if (starts) starts[n] = all.n;
    if (n_contours) *n_contours = n;
    if (n_points) *n_points = all.n;
    if (n == 0) {
        if (offs && cap_contours >= 0) offs[0] = 0;
        goto done;
    }
    if (n > cap_contours || all.n > cap_pts || !pts || !offs) {
        rc = -2; /* counts above are the required capacities */
        goto done;
    }
    /* reverse discovery order */
    {
        int o = 0;
        for (int k = n - 1; k >= 0; k--) {
            int len = starts[k + 1] - starts[k];
            offs[n - 1 - k] = o;
            memcpy(pts + o, all.p + starts[k], (size_t)len * sizeof(orc_point));
            o += len;
        }
        offs[n] = o;
    }
done:
    free(img);
    free(all.p);
    free(starts);
    return rc;
}

/* ------------------------------------------------- objdetect.cpp:64 contourArea */
/* [OCV] cv::contourArea(contour, oriented=false): points as float, shoelace in double. */
double orc_contour_area(const orc_point* pts, int n)
{
    if (n == 0) return 0.0;
    double a00 = 0;
    float px = (float)pts[n - 1].x, py = (float)pts[n - 1].y;
    for (int i = 0; i < n; i++) {
        float x = (float)pts[i].x, y = (float)pts[i].y;
        a00 += (double)px * y - (double)py * x;
        px = x;
        py = y;
    }
    a00 *= 0.5;
    return fabs(a00);
}

/* ------------------------------------------------- objdetect.cpp:68 fitEllipseDirect */

/* [OCV] cv::eigenNonSymmetric for a 3x3 real matrix: JAMA EigenvalueDecomposition
 * (orthes + hqr2, public-domain NIST algorithm as adopted by modules/core/src/lda.cpp).
 * V columns are the eigenvectors, d/e the real/imaginary eigenvalue parts. */
#define EN 3
static void cdiv_(double xr, double xi, double yr, double yi, double* cr, double* ci)
{
    double r, d;
    if (fabs(yr) > fabs(yi)) {
        r = yi / yr;
        d = yr + r * yi;
        *cr = (xr + r * xi) / d;
        *ci = (xi - r * xr) / d;
    } else {
        r = yr / yi;
        d = yi + r * yr;
        *cr = (r * xr + xi) / d;
        *ci = (r * xi - xr) / d;
    }
}

static void eig_orthes(double H[EN][EN], double V[EN][EN])
{
    double ort[EN];
    const int low = 0, high = EN - 1;
    for (int m = low + 1; m <= high - 1; m++) {
        double scale = 0.0;
        for (int i = m; i <= high; i++) scale = scale + fabs(H[i][m - 1]);
        if (scale != 0.0) {
            double h = 0.0;
            for (int i = high; i >= m; i--) {
                ort[i] = H[i][m - 1] / scale;
                h += ort[i] * ort[i];
            }
            double g = sqrt(h);
            if (ort[m] > 0) g = -g;
            h = h - ort[m] * g;
            ort[m] = ort[m] - g;
            for (int j = m; j < EN; j++) {
                double f = 0.0;
                for (int i = high; i >= m; i--) f += ort[i] * H[i][j];
                f = f / h;
                for (int i = m; i <= high; i++) H[i][j] -= f * ort[i];
            }
            for (int i = 0; i <= high; i++) {
                double f = 0.0;
                for (int j = high; j >= m; j--) f += ort[j] * H[i][j];
                f = f / h;
                for (int j = m; j <= high; j++) H[i][j] -= f * ort[j];
            }
            ort[m] = scale * ort[m];
            H[m][m - 1] = scale * g;
        }
    }
    for (int i = 0; i < EN; i++)
        for (int j = 0; j < EN; j++) V[i][j] = (i == j ? 1.0 : 0.0);
    for (int m = high - 1; m >= low + 1; m--) {
        if (H[m][m - 1] != 0.0) {
            for (int i = m + 1; i <= high; i++) ort[i] = H[i][m - 1];
            for (int j = m; j <= high; j++) {
                double g = 0.0;
                for (int i = m; i <= high; i++) g += ort[i] * V[i][j];
                g = (g / ort[m]) / H[m][m - 1];
                for (int i = m; i <= high; i++) V[i][j] += g * ort[i];
            }
        }
    }
}

static void eig_hqr2(double H[EN][EN], double V[EN][EN], double d[EN], double e[EN])
{
    const int nn = EN;
    int n = nn - 1;
    const int low = 0, high = nn - 1;
    const double eps = 2.220446049250313e-16; /* pow(2.0, -52.0) */
    double exshift = 0.0;
    double p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
    double norm = 0.0;
    for (int i = 0; i < nn; i++)
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < nn; j++) norm = norm + fabs(H[i][j]);

    int iter = 0;
    while (n >= low) {
        int l = n;
        while (l > low) {
            s = fabs(H[l - 1][l - 1]) + fabs(H[l][l]);
            if (s == 0.0) s = norm;
            if (fabs(H[l][l - 1]) < eps * s) break;
            l--;
        }
        if (l == n) { /* one root */
            H[n][n] = H[n][n] + exshift;
            d[n] = H[n][n];
            e[n] = 0.0;
            n--;
            iter = 0;
        } else if (l == n - 1) { /* two roots */
            w = H[n][n - 1] * H[n - 1][n];
            p = (H[n - 1][n - 1] - H[n][n]) / 2.0;
            q = p * p + w;
            z = sqrt(fabs(q));
            H[n][n] = H[n][n] + exshift;
            H[n - 1][n - 1] = H[n - 1][n - 1] + exshift;
            x = H[n][n];
            if (q >= 0) { /* real pair */
                if (p >= 0) z = p + z; else z = p - z;
                d[n - 1] = x + z;
                d[n] = d[n - 1];
                if (z != 0.0) d[n] = x - w / z;
                e[n - 1] = 0.0;
                e[n] = 0.0;
                x = H[n][n - 1];
                s = fabs(x) + fabs(z);
                p = x / s;
                q = z / s;
                r = sqrt(p * p + q * q);
                p = p / r;
                q = q / r;
                for (int j = n - 1; j < nn; j++) {
                    z = H[n - 1][j];
                    H[n - 1][j] = q * z + p * H[n][j];
                    H[n][j] = q * H[n][j] - p * z;
                }
                for (int i = 0; i <= n; i++) {
                    z = H[i][n - 1];
                    H[i][n - 1] = q * z + p * H[i][n];
                    H[i][n] = q * H[i][n] - p * z;
                }
                for (int i = low; i <= high; i++) {
                    z = V[i][n - 1];
                    V[i][n - 1] = q * z + p * V[i][n];
                    V[i][n] = q * V[i][n] - p * z;
                }
            } else { /* complex pair */
                d[n - 1] = x + p;
                d[n] = x + p;
                e[n - 1] = z;
                e[n] = -z;
            }
            n = n - 2;
            iter = 0;
        } else { /* no convergence yet */
            x = H[n][n];
            y = 0.0;
            w = 0.0;
            if (l < n) {
                y = H[n - 1][n - 1];
                w = H[n][n - 1] * H[n - 1][n];
            }
            if (iter == 10) { /* Wilkinson's original ad hoc shift */
                exshift += x;
                for (int i = low; i <= n; i++) H[i][i] -= x;
                s = fabs(H[n][n - 1]) + fabs(H[n - 1][n - 2]);
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            if (iter == 30) { /* MATLAB's new ad hoc shift */
                s = (y - x) / 2.0;
                s = s * s + w;
                if (s > 0) {
                    s = sqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / 2.0 + s);
                    for (int i = low; i <= n; i++) H[i][i] -= s;
                    exshift += s;
                    x = y = w = 0.964;
                }
            }
            iter = iter + 1;
            if (iter > 300) { /* not in JAMA: a guard so the oracle (and the GPU twin) always terminate */
                d[n] = H[n][n] + exshift;
                e[n] = 0.0;
                n--;
                iter = 0;
                continue;
            }
            int m = n - 2;
            while (m >= l) {
                z = H[m][m];
                r = x - z;
                s = y - z;
                p = (r * s - w) / H[m + 1][m] + H[m][m + 1];
                q = H[m + 1][m + 1] - z - r - s;
                r = H[m + 2][m + 1];
                s = fabs(p) + fabs(q) + fabs(r);
                p = p / s;
                q = q / s;
                r = r / s;
                if (m == l) break;
                if (fabs(H[m][m - 1]) * (fabs(q) + fabs(r)) <
                    eps * (fabs(p) * (fabs(H[m - 1][m - 1]) + fabs(z) + fabs(H[m + 1][m + 1]))))
                    break;
                m--;
            }
            for (int i = m + 2; i <= n; i++) {
                H[i][i - 2] = 0.0;
                if (i > m + 2) H[i][i - 3] = 0.0;
            }
            for (int k = m; k <= n - 1; k++) {
                int notlast = (k != n - 1);
                if (k != m) {
                    p = H[k][k - 1];
                    q = H[k + 1][k - 1];
                    r = (notlast ? H[k + 2][k - 1] : 0.0);
                    x = fabs(p) + fabs(q) + fabs(r);
                    if (x != 0.0) {
                        p = p / x;
                        q = q / x;
                        r = r / x;
                    }
                }
                if (x == 0.0) break;
                s = sqrt(p * p + q * q + r * r);
                if (p < 0) s = -s;
                if (s != 0) {
                    if (k != m) H[k][k - 1] = -s * x;
                    else if (l != m) H[k][k - 1] = -H[k][k - 1];
                    p = p + s;
                    x = p / s;
                    y = q / s;
                    z = r / s;
                    q = q / p;
                    r = r / p;
                    for (int j = k; j < nn; j++) {
                        p = H[k][j] + q * H[k + 1][j];
                        if (notlast) {
                            p = p + r * H[k + 2][j];
                            H[k + 2][j] = H[k + 2][j] - p * z;
                        }
                        H[k][j] = H[k][j] - p * x;
                        H[k + 1][j] = H[k + 1][j] - p * y;
                    }
                    int imax = (n < k + 3 ? n : k + 3);
                    for (int i = 0; i <= imax; i++) {
                        p = x * H[i][k] + y * H[i][k + 1];
                        if (notlast) {
                            p = p + z * H[i][k + 2];
                            H[i][k + 2] = H[i][k + 2] - p * r;
                        }
                        H[i][k] = H[i][k] - p;
                        H[i][k + 1] = H[i][k + 1] - p * q;
                    }
                    for (int i = low; i <= high; i++) {
                        p = x * V[i][k] + y * V[i][k + 1];
                        if (notlast) {
                            p = p + z * V[i][k + 2];
                            V[i][k + 2] = V[i][k + 2] - p * r;
                        }
                        V[i][k] = V[i][k] - p;
                        V[i][k + 1] = V[i][k + 1] - p * q;
                    }
                }
            }
        }
    }

    if (norm == 0.0) return;

    for (n = nn - 1; n >= 0; n--) {
        p = d[n];
        q = e[n];
        if (q == 0) { /* real vector */
            int l = n;
            H[n][n] = 1.0;
            for (int i = n - 1; i >= 0; i--) {
                w = H[i][i] - p;
                r = 0.0;
                for (int j = l; j <= n; j++) r = r + H[i][j] * H[j][n];
                if (e[i] < 0.0) {
                    z = w;
                    s = r;
                } else {
                    l = i;
                    if (e[i] == 0.0) {
                        if (w != 0.0) H[i][n] = -r / w;
                        else H[i][n] = -r / (eps * norm);
                    } else {
                        x = H[i][i + 1];
                        y = H[i + 1][i];
                        q = (d[i] - p) * (d[i] - p) + e[i] * e[i];
                        t = (x * s - z * r) / q;
                        H[i][n] = t;
                        if (fabs(x) > fabs(z)) H[i + 1][n] = (-r - w * t) / x;
                        else H[i + 1][n] = (-s - y * t) / z;
                    }
                    t = fabs(H[i][n]);
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; j++) H[j][n] = H[j][n] / t;
                }
            }
        } else if (q < 0) { /* complex vector */
            int l = n - 1;
            double cr, ci;
            if (fabs(H[n][n - 1]) > fabs(H[n - 1][n])) {
                H[n - 1][n - 1] = q / H[n][n - 1];
                H[n - 1][n] = -(H[n][n] - p) / H[n][n - 1];
            } else {
                cdiv_(0.0, -H[n - 1][n], H[n - 1][n - 1] - p, q, &cr, &ci);
                H[n - 1][n - 1] = cr;
                H[n - 1][n] = ci;
            }
            H[n][n - 1] = 0.0;
            H[n][n] = 1.0;
            for (int i = n - 2; i >= 0; i--) {
                double ra = 0.0, sa = 0.0, vr, vi;
                for (int j = l; j <= n; j++) {
                    ra = ra + H[i][j] * H[j][n - 1];
                    sa = sa + H[i][j] * H[j][n];
                }
                w = H[i][i] - p;
                if (e[i] < 0.0) {
                    z = w;
                    r = ra;
                    s = sa;
                } else {
                    l = i;
                    if (e[i] == 0) {
                        cdiv_(-ra, -sa, w, q, &cr, &ci);
                        H[i][n - 1] = cr;
                        H[i][n] = ci;
                    } else {
                        x = H[i][i + 1];
                        y = H[i + 1][i];
                        vr = (d[i] - p) * (d[i] - p) + e[i] * e[i] - q * q;
                        vi = (d[i] - p) * 2.0 * q;
                        if (vr == 0.0 && vi == 0.0)
                            vr = eps * norm * (fabs(w) + fabs(q) + fabs(x) + fabs(y) + fabs(z));
                        cdiv_(x * r - z * ra + q * sa, x * s - z * sa - q * ra, vr, vi, &cr, &ci);
                        H[i][n - 1] = cr;
                        H[i][n] = ci;
                        if (fabs(x) > (fabs(z) + fabs(q))) {
                            H[i + 1][n - 1] = (-ra - w * H[i][n - 1] + q * H[i][n]) / x;
                            H[i + 1][n] = (-sa - w * H[i][n] - q * H[i][n - 1]) / x;
                        } else {
                            cdiv_(-r - y * H[i][n - 1], -s - y * H[i][n], z, q, &cr, &ci);
                            H[i + 1][n - 1] = cr;
                            H[i + 1][n] = ci;
                        }
                    }
                    t = fabs(H[i][n - 1]) > fabs(H[i][n]) ? fabs(H[i][n - 1]) : fabs(H[i][n]);
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; j++) {
                            H[j][n - 1] = H[j][n - 1] / t;
                            H[j][n] = H[j][n] / t;
                        }
                }
            }
        }
    }
    for (int j = nn - 1; j >= low; j--)
        for (int i = low; i <= high; i++) {
            z = 0.0;
            int kmax = (j < high ? j : high);
            for (int k = low; k <= kmax; k++) z = z + V[i][k] * H[k][j];
            V[i][j] = z;
        }
}

/* [OCV] eigenNonSymmetric: eigenvalues sorted descending, eigenvectors as rows. */
static void eigen_nonsymmetric3(const double M[3][3], double eval[3], double evec[3][3])
{
    double H[3][3], V[3][3], d[3] = {0, 0, 0}, e[3] = {0, 0, 0};
    memcpy(H, M, sizeof(H));
    eig_orthes(H, V);
    eig_hqr2(H, V, d, e);
    int idx[3] = {0, 1, 2};
    for (int i = 1; i < 3; i++) { /* stable insertion sort, descending */
        int k = idx[i], j = i - 1;
        while (j >= 0 && d[idx[j]] < d[k]) {
            idx[j + 1] = idx[j];
            j--;
        }
        idx[j + 1] = k;
    }
    for (int i = 0; i < 3; i++) {
        eval[i] = d[idx[i]];
        for (int j = 0; j < 3; j++) evec[i][j] = V[j][idx[i]];
    }
}

/* Cyclic Jacobi for a symmetric k x k matrix (k <= 5).  BUILD-DEFINED: OpenCV's general fit
 * solves its least-squares systems with a Jacobi SVD of the n x k design matrix; this oracle
 * solves the same systems through the k x k normal equations (sequential sums in point order)
 * and this eigen-solver, so that the per-contour work is one pass over the points.  Part of
 * the "parity unpinned" surface. */
static void jacobi_sym(double* A, int k, double* lam, double* V)
{
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) V[i * k + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < k; p++)
            for (int q = p + 1; q < k; q++) off += fabs(A[p * k + q]);
        if (off == 0.0) break;
        for (int p = 0; p < k; p++)
            for (int q = p + 1; q < k; q++) {
                double apq = A[p * k + q];
                if (apq == 0.0) continue;
                double app = A[p * k + p], aqq = A[q * k + q];
                if (fabs(apq) < 1e-300 || fabs(apq) <= 1.1102230246251565e-16 * 1e-3 * sqrt(fabs(app * aqq))) {
                    A[p * k + q] = A[q * k + p] = 0.0;
                    continue;
                }
                double theta = (aqq - app) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (theta < 0) t = -t;
                double c = 1.0 / sqrt(t * t + 1.0);
                double s = t * c;
                A[p * k + p] = app - t * apq;
                A[q * k + q] = aqq + t * apq;
                A[p * k + q] = A[q * k + p] = 0.0;
                for (int r = 0; r < k; r++) {
                    if (r != p && r != q) {
                        double arp = A[r * k + p], arq = A[r * k + q];
                        double nrp = c * arp - s * arq;
                        double nrq = s * arp + c * arq;
                        A[r * k + p] = A[p * k + r] = nrp;
                        A[r * k + q] = A[q * k + r] = nrq;
                    }
                    double vrp = V[r * k + p], vrq = V[r * k + q];
                    V[r * k + p] = c * vrp - s * vrq;
                    V[r * k + q] = s * vrp + c * vrq;
                }
            }
    }
    for (int i = 0; i < k; i++) lam[i] = A[i * k + i];
}

/* Least squares min |A x - b| for an n x k design matrix (k <= 5) that is never stored: rows are produced on demand.
 * BUILD-DEFINED (see jacobi_sym): OpenCV runs a Jacobi SVD on the n x k matrix; here
 *   1. G = A^T A and g = A^T b by sequential sums in point order (the caller), x0 = pinv(G) g through the Jacobi
 *      eigen-decomposition G = V diag(lam) V^T  (normal_factor / normal_apply);
 *   2. two steps of iterative refinement  x += pinv(G) A^T (b - A x)  with the residual taken from the rows themselves.
 * Step 1 alone squares the condition number: on thin bars (aspect > 30, exactly the contours that reach this fallback, and
 * rm::filter_lightblobs admits ratios up to 80) cond(A) is 1e4..1e6 and x0 is good to 1e-8..1e-5 only, which moved one
 * float32 output in fifty by an ulp against an SVD solve.  Every refinement step multiplies the error by cond^2 * eps, so
 * two steps land on the accuracy of a backward-stable solver (1e-12..1e-16, tests/test_oracle_general_fit_svd.py compares
 * with LAPACK's SVD least squares).  The refinement sums are order dependent and defined wave-shaped, the way the device
 * computes them: 64 partial sums over the points i = l, l + 64, ... in increasing i, reduced by a butterfly (strides
 * 32, 16, 8, 4, 2, 1; v[l] += v[l ^ stride]). */
typedef struct { int k; double lam[5], V[25], w[5], thr, wmax, wmin; } normal_fac;

static void normal_factor(const double* G, int k, normal_fac* F)
{
    double A[25];
    memcpy(A, G, sizeof(double) * k * k);
    F->k = k;
    jacobi_sym(A, k, F->lam, F->V);
    double wsum = 0, mx = 0, mn = 0;
    for (int i = 0; i < k; i++) {
        F->w[i] = F->lam[i] > 0 ? sqrt(F->lam[i]) : 0.0;
        wsum += F->w[i];
        if (i == 0 || F->w[i] > mx) mx = F->w[i];
        if (i == 0 || F->w[i] < mn) mn = F->w[i];
    }
    F->thr = 2.0 * DBL_EPSILON * wsum; /* [OCV] SVBackSubst threshold */
    F->wmax = mx; /* extreme singular values of A (sqrt of G's eigenvalues) */
    F->wmin = mn;
}

static void normal_apply(const normal_fac* F, const double* g, double* x)
{
    const int k = F->k;
    for (int i = 0; i < k; i++) x[i] = 0.0;
    for (int c = 0; c < k; c++) {
        if (!(F->w[c] > F->thr)) continue;
        double dot = 0.0;
        for (int r = 0; r < k; r++) dot += F->V[r * k + c] * g[r];
        dot = dot / F->lam[c];
        for (int r = 0; r < k; r++) x[r] += dot * F->V[r * k + c];
    }
}

static double butterfly64(double v[64])
{
    for (int d = 32; d >= 1; d >>= 1) {
        double t[64];
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ d];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}

typedef void (*row_fn)(const void* ctx, int i, double* row);
#define NORMAL_REFINE_STEPS 2
static void normal_refine(const normal_fac* F, row_fn make_row, const void* ctx, int n, double bconst, double* x)
{
    const int k = F->k;
    for (int step = 0; step < NORMAL_REFINE_STEPS; step++) {
        double part[5][64];
        memset(part, 0, sizeof(part));
        for (int l = 0; l < 64; l++)
            for (int i = l; i < n; i += 64) {
                double row[5];
                make_row(ctx, i, row);
                double t = row[0] * x[0];
                for (int a = 1; a < k; a++) t += row[a] * x[a];
                const double r = bconst - t;
                for (int a = 0; a < k; a++) part[a][l] += row[a] * r;
            }
        double h[5], dx[5];
        for (int a = 0; a < k; a++) h[a] = butterfly64(part[a]);
        normal_apply(F, h, dx);
        for (int a = 0; a < k; a++) x[a] += dx[a];
    }
}

static void get_ofs(int i, float eps, float* ox, float* oy)
{ /* [OCV] getOfs */
    *ox = (float)(((i & 1) * 2 - 1)) * eps;
    *oy = (float)(((i & 2) - 1)) * eps;
}

/* G = A^T A (k x k, both triangles filled) and g = A^T (bconst, ..., bconst) of the n x k design matrix given by make_row, summed
 * the way the device sums them: 64 partial sums over the points i = l, l + 64, ... in increasing i, then the butterfly.  (The
 * direct fit's scatter matrix keeps OpenCV's sequential order; this matrix only seeds the refinement, whose fixed point does
 * not depend on its rounding.) */
static void normal_equations(row_fn make_row, const void* ctx, int n, int k, double bconst, double* G, double* g)
{
    double pg[25][64], pv[5][64];
    memset(pg, 0, sizeof(pg));
    memset(pv, 0, sizeof(pv));
    for (int l = 0; l < 64; l++)
        for (int i = l; i < n; i += 64) {
            double row[5];
            make_row(ctx, i, row);
            for (int a = 0; a < k; a++) {
                for (int b = a; b < k; b++) pg[a * k + b][l] += row[a] * row[b];
                pv[a][l] += row[a] * bconst;
            }
        }
    for (int a = 0; a < k; a++) {
        for (int b = a; b < k; b++) G[a * k + b] = G[b * k + a] = butterfly64(pg[a * k + b]);
        g[a] = butterfly64(pv[a]);
    }
}

/* design-matrix rows of the general fit (the n x 5 and n x 3 matrices of [OCV] fitEllipseNoDirect), produced on demand */
typedef struct { const orc_point* pts; float cx, cy; double scale; float eps; double r0, r1; } gen_rows;
static void gen_pxy(const gen_rows* R, int i, double* px, double* py)
{
    float ox = 0, oy = 0;
    if (R->eps != 0.0f) get_ofs(i, R->eps, &ox, &oy);
    const float fx = ((float)R->pts[i].x + ox) - R->cx, fy = ((float)R->pts[i].y + oy) - R->cy;
    *px = fx * R->scale;
    *py = fy * R->scale;
}
static void gen_row5(const void* ctx, int i, double* row)
{
    double px, py;
    gen_pxy((const gen_rows*)ctx, i, &px, &py);
    row[0] = -px * px; row[1] = -py * py; row[2] = -px * py; row[3] = px; row[4] = py;
}
static void gen_row3(const void* ctx, int i, double* row)
{
    const gen_rows* R = (const gen_rows*)ctx;
    double px, py;
    gen_pxy(R, i, &px, &py);
    row[0] = (px - R->r0) * (px - R->r0); row[1] = (py - R->r1) * (py - R->r1); row[2] = (px - R->r0) * (py - R->r1);
}

/* [OCV] fitEllipseNoDirect (general "LIN" conic fit, D. Weiss): structure restated, the two
 * least-squares sub-problems solved via normal_factor / normal_apply / normal_refine (BUILD-DEFINED, see above). */
static void fit_ellipse_general(const orc_point* pts, int n, orc_rrect* box)
{
    const double min_eps = 1e-8;
    float cx = 0, cy = 0; /* Point2f accumulator */
    for (int i = 0; i < n; i++) {
        cx += (float)pts[i].x;
        cy += (float)pts[i].y;
    }
    cx /= (float)n;
    cy /= (float)n;
    double s = 0;
    for (int i = 0; i < n; i++) {
        float px = (float)pts[i].x - cx, py = (float)pts[i].y - cy;
        s += fabs((double)px) + fabs((double)py);
    }
    double scale = 100.0 / (s > FLT_EPSILON ? s : (double)FLT_EPSILON);
    double gfp[5], rp[5] = {0, 0, 0, 0, 0};
    float eps = 0.0f;
    gen_rows R = {pts, cx, cy, scale, 0.0f, 0.0, 0.0};
    for (int iter = 0; iter < 2; iter++) {
        double G[25], g[5];
        normal_fac F;
        R.eps = iter ? eps : 0.0f;
        normal_equations(gen_row5, &R, n, 5, 10000.0, G, g);
        normal_factor(G, 5, &F);
        if (iter == 0 && F.wmax * FLT_EPSILON > F.wmin) {
            eps = (float)(s / (n * 2) * 1e-3);
            continue;
        }
        normal_apply(&F, g, gfp);
        normal_refine(&F, gen_row5, &R, n, 10000.0, gfp);
        break;
    }
    /* centre: differentiate the general form */
    {
        double a00 = 2 * gfp[0], a01 = gfp[2], a11 = 2 * gfp[1];
        double det = a00 * a11 - a01 * a01;
        if (det != 0.0) {
            rp[0] = (gfp[3] * a11 - gfp[4] * a01) / det;
            rp[1] = (a00 * gfp[4] - a01 * gfp[3]) / det;
        }
    }
    /* re-fit A..C with that centre */
    {
        double G[9], g[3];
        normal_fac F;
        R.eps = eps;
        R.r0 = rp[0];
        R.r1 = rp[1];
        normal_equations(gen_row3, &R, n, 3, 1.0, G, g);
        normal_factor(G, 3, &F);
        normal_apply(&F, g, gfp);
        normal_refine(&F, gen_row3, &R, n, 1.0, gfp);
    }
    double t;
    rp[4] = -0.5 * m_atan2(gfp[2], gfp[1] - gfp[0]);
    if (fabs(gfp[2]) > min_eps) t = gfp[2] / m_sin(-2.0 * rp[4]);
    else t = gfp[1] - gfp[0];
    rp[2] = fabs(gfp[0] + gfp[1] - t);
    if (rp[2] > min_eps) rp[2] = sqrt(2.0 / rp[2]);
    rp[3] = fabs(gfp[0] + gfp[1] + t);
    if (rp[3] > min_eps) rp[3] = sqrt(2.0 / rp[3]);
    box->cx = (float)(rp[0] / scale) + cx;
    box->cy = (float)(rp[1] / scale) + cy;
    box->w = (float)(rp[2] * 2 / scale);
    box->h = (float)(rp[3] * 2 / scale);
    box->angle = 0.0f; /* [OCV] only the swapped branch assigns the angle; the unswapped case arises only
                          for an axis-aligned ellipse with a vertical major axis, where 0 is the answer */
    if (box->w > box->h) {
        float tmp = box->w;
        box->w = box->h;
        box->h = tmp;
        box->angle = (float)(90 + rp[4] * 180 / ORC_PI);
    }
    if (box->angle < -180) box->angle += 360;
    if (box->angle > 360) box->angle -= 360;
}

/* [OCV] isGoodBox: the direct/AMS fits are replaced by the general fit when the box is wild */
static double g_dbg_det[2];
double orc_debug_last_det(int k) { return g_dbg_det[k & 1]; }
static int is_good_box(const orc_rrect* b) { return (b->h <= b->w * 30) && (b->w <= b->h * 30); }

/* [OCV] cv::fitEllipseDirect (Fitzgibbon et al. 1999, numerically stable form of Halir &
 * Flusser): centroid -> L1 scale -> n x 6 design rows [x^2 xy y^2 x y 1] -> DM = A^T A / n
 * (each entry a sequential sum in point order) -> reduced 3x3 system -> eigenvector with
 * 4ac-b^2 > 0 -> conic -> RotatedRect (width <= height, angle in [0,180)). */
int orc_fit_ellipse_direct(const orc_point* pts, int n, orc_rrect* box)
{
    double cx = 0, cy = 0;
    for (int i = 0; i < n; i++) {
        cx += (float)pts[i].x;
        cy += (float)pts[i].y;
    }
    cx /= n;
    cy /= n;
    double s = 0;
    for (int i = 0; i < n; i++) s += fabs((float)pts[i].x - cx) + fabs((float)pts[i].y - cy);
    double scale = 100.0 / (s > FLT_EPSILON ? s : (double)FLT_EPSILON);

    double DM[6][6], TM[3][3], M[3][3], Ts = 0;
    float eps = 0;
    int iter;
    for (iter = 0; iter < 2; iter++) {
        double acc[6][6];
        memset(acc, 0, sizeof(acc));
        for (int i = 0; i < n; i++) {
            float ox, oy;
            get_ofs(i, eps, &ox, &oy);
            double px = (((float)pts[i].x + ox) - cx) * scale, py = (((float)pts[i].y + oy) - cy) * scale;
            double row[6] = {px * px, px * py, py * py, px, py, 1.0};
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++) acc[a][b] += row[a] * row[b]; /* mulTransposed: sequential over rows */
        }
        double inv_n = 1.0 / n;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) DM[a][b] = DM[b][a] = acc[a][b] * inv_n; /* DM *= (1.0/n) */

        /* TM = -adj(S3) * S2^T, written as the six-term cofactor expressions */
        for (int c = 0; c < 3; c++) {
            TM[0][c] = DM[c][5] * DM[3][5] * DM[4][4] - DM[c][5] * DM[3][4] * DM[4][5] - DM[c][4] * DM[3][5] * DM[5][4] +
                       DM[c][3] * DM[4][5] * DM[5][4] + DM[c][4] * DM[3][4] * DM[5][5] - DM[c][3] * DM[4][4] * DM[5][5];
            TM[1][c] = DM[c][5] * DM[3][3] * DM[4][5] - DM[c][5] * DM[3][5] * DM[4][3] + DM[c][4] * DM[3][5] * DM[5][3] -
                       DM[c][3] * DM[4][5] * DM[5][3] - DM[c][4] * DM[3][3] * DM[5][5] + DM[c][3] * DM[4][3] * DM[5][5];
            TM[2][c] = DM[c][5] * DM[3][4] * DM[4][3] - DM[c][5] * DM[3][3] * DM[4][4] - DM[c][4] * DM[3][4] * DM[5][3] +
                       DM[c][3] * DM[4][4] * DM[5][3] + DM[c][4] * DM[3][3] * DM[5][4] - DM[c][3] * DM[4][3] * DM[5][4];
        }
        Ts = (-(DM[3][5] * DM[4][4] * DM[5][3]) + DM[3][4] * DM[4][5] * DM[5][3] + DM[3][5] * DM[4][3] * DM[5][4] -
              DM[3][3] * DM[4][5] * DM[5][4] - DM[3][4] * DM[4][3] * DM[5][5] + DM[3][3] * DM[4][4] * DM[5][5]);
        for (int c = 0; c < 3; c++) {
            M[0][c] = (DM[2][c] + (DM[2][3] * TM[0][c] + DM[2][4] * TM[1][c] + DM[2][5] * TM[2][c]) / Ts) / 2.;
            M[1][c] = -DM[1][c] - (DM[1][3] * TM[0][c] + DM[1][4] * TM[1][c] + DM[1][5] * TM[2][c]) / Ts;
            M[2][c] = (DM[0][c] + (DM[0][3] * TM[0][c] + DM[0][4] * TM[1][c] + DM[0][5] * TM[2][c]) / Ts) / 2.;
        }
        double det = M[0][0] * (M[1][1] * M[2][2] - M[2][1] * M[1][2]) - M[0][1] * (M[1][0] * M[2][2] - M[2][0] * M[1][2]) +
                     M[0][2] * (M[1][0] * M[2][1] - M[2][0] * M[1][1]);
        g_dbg_det[iter] = det;
        if (fabs(det) > 1.0e-10) break;
        eps = (float)(s / (n * 2) * 1e-2);
    }
    if (iter < 2) {
        double eval[3], ev[3][3], cond[3];
        int i;
        eigen_nonsymmetric3(M, eval, ev);
        cond[0] = (4.0 * ev[0][0] * ev[0][2] - ev[0][1] * ev[0][1]);
        cond[1] = (4.0 * ev[1][0] * ev[1][2] - ev[1][1] * ev[1][1]);
        cond[2] = (4.0 * ev[2][0] * ev[2][2] - ev[2][1] * ev[2][1]);
        if (cond[0] < cond[1]) i = (cond[1] < cond[2]) ? 2 : 1;
        else i = (cond[0] < cond[2]) ? 2 : 0;
        double norm = sqrt(ev[i][0] * ev[i][0] + ev[i][1] * ev[i][1] + ev[i][2] * ev[i][2]);
        if (((ev[i][0] < 0.0 ? -1 : 1) * (ev[i][1] < 0.0 ? -1 : 1) * (ev[i][2] < 0.0 ? -1 : 1)) <= 0.0) norm = -1.0 * norm;
        double pv0 = ev[i][0] / norm, pv1 = ev[i][1] / norm, pv2 = ev[i][2] / norm;
        /* Q = (TM . pVec) / Ts */
        double q0 = (TM[0][0] * pv0 + TM[0][1] * pv1 + TM[0][2] * pv2) / Ts;
        double q1 = (TM[1][0] * pv0 + TM[1][1] * pv1 + TM[1][2] * pv2) / Ts;
        double q2 = (TM[2][0] * pv0 + TM[2][1] * pv1 + TM[2][2] * pv2) / Ts;
        double u1 = pv2 * q0 * q0 - pv1 * q0 * q1 + pv0 * q1 * q1 + pv1 * pv1 * q2;
        double u2 = pv0 * pv2 * q2;
        double l1 = sqrt(pv1 * pv1 + (pv0 - pv2) * (pv0 - pv2));
        double l2 = pv0 + pv2;
        double l3 = pv1 * pv1 - 4 * pv0 * pv2;
        double p1 = 2 * pv2 * q0 - pv1 * q1;
        double p2 = 2 * pv0 * q1 - pv1 * q0;
        double x0 = (p1 / l3 / scale) + cx;
        double y0 = (p2 / l3 / scale) + cy;
        double a = sqrt(2.) * sqrt((u1 - 4.0 * u2) / ((l1 - l2) * l3)) / scale;
        double b = sqrt(2.) * sqrt(-1.0 * ((u1 - 4.0 * u2) / ((l1 + l2) * l3))) / scale;
        double theta;
        if (pv1 == 0) theta = (pv0 < pv2) ? 0 : ORC_PI / 2.;
        else theta = ORC_PI / 2. + 0.5 * m_atan2(pv1, (pv0 - pv2));
        box->cx = (float)x0;
        box->cy = (float)y0;
        box->w = (float)(2.0 * a);
        box->h = (float)(2.0 * b);
        if (box->w > box->h) {
            float tmp = box->w;
            box->w = box->h;
            box->h = tmp;
            box->angle = (float)(m_fmod180(90 + theta * 180 / ORC_PI));
        } else {
            box->angle = (float)(m_fmod180(theta * 180 / ORC_PI));
        }
        if (is_good_box(box)) return 0;
    }
    fit_ellipse_general(pts, n, box);
    return 1;
}

/* ------------------------------------------------- core.cpp:265-283, 9-19 */

/* [OCV] cv::RotatedRect::points */
void orc_rrect_points(const orc_rrect* r, float pt[4][2])
{
    double ang = r->angle * ORC_PI / 180.;
    float b = (float)m_cos(ang) * 0.5f;
    float a = (float)m_sin(ang) * 0.5f;
    pt[0][0] = r->cx - a * r->h - b * r->w;
    pt[0][1] = r->cy + b * r->h - a * r->w;
    pt[1][0] = r->cx + a * r->h - b * r->w;
    pt[1][1] = r->cy - b * r->h - a * r->w;
    pt[2][0] = 2 * r->cx - pt[0][0];
    pt[2][1] = 2 * r->cy - pt[0][1];
    pt[3][0] = 2 * r->cx - pt[1][0];
    pt[3][1] = 2 * r->cy - pt[1][1];
}

void orc_make_lightblob(const orc_rrect* box, int camp, orc_lightblob* out)
{
    /* core.cpp:9-14 */
    out->angle = box->angle > 90 ? box->angle - 90 : box->angle + 90;
    out->target = camp;
    out->center[0] = box->cx;
    out->center[1] = box->cy;
    /* core.cpp:265-283 reorder_vertices(RECT_TALL): points(), std::sort by y (4 elements =
     * insertion sort, ties keep order), then left/right by x. */
    float t[4][2];
    orc_rrect_points(box, t);
    for (int i = 1; i < 4; i++) {
        float kx = t[i][0], ky = t[i][1];
        int j = i - 1;
        while (j >= 0 && ky < t[j][1]) {
            t[j + 1][0] = t[j][0];
            t[j + 1][1] = t[j][1];
            j--;
        }
        t[j + 1][0] = kx;
        t[j + 1][1] = ky;
    }
    int swap_up = t[0][0] < t[1][0], swap_down = t[2][0] < t[3][0];
    const float* o0 = swap_down ? t[2] : t[3];
    const float* o1 = swap_up ? t[0] : t[1];
    const float* o2 = swap_up ? t[1] : t[0];
    const float* o3 = swap_down ? t[3] : t[2];
    out->vertices[0][0] = o0[0]; out->vertices[0][1] = o0[1];
    out->vertices[1][0] = o1[0]; out->vertices[1][1] = o1[1];
    out->vertices[2][0] = o2[0]; out->vertices[2][1] = o2[1];
    out->vertices[3][0] = o3[0]; out->vertices[3][1] = o3[1];
    /* core.cpp:18 */
    out->size[0] = box->h < box->w ? box->h : box->w;
    out->size[1] = box->h < box->w ? box->w : box->h;
}

/* ------------------------------------------------- core.cpp:285-404 helpers */
static float point_distance(const float a[2], const float b[2])
{ /* core.cpp:285-288: float differences, pow(.,2)+pow(.,2) and sqrt in double */
    double dx = (double)(a[0] - b[0]), dy = (double)(a[1] - b[1]);
    return (float)sqrt(dx * dx + dy * dy);
}

static void extend_cord(const float pt1[2], const float pt2[2], float deltaLen, float dst1[2], float dst2[2])
{ /* core.cpp:295-380; unqualified abs/atan2/sin/cos on floats: float overloads unless orc_set_overload_mode says otherwise */
    if (pt1[0] == pt2[0]) {
        dst1[0] = pt1[0];
        dst2[0] = pt1[0];
        if (pt1[1] > pt2[1]) { dst1[1] = pt1[1] + deltaLen; dst2[1] = pt2[1] - deltaLen; }
        else                 { dst1[1] = pt1[1] - deltaLen; dst2[1] = pt2[1] + deltaLen; }
    } else if (pt1[1] == pt2[1]) {
        dst1[1] = pt1[1];
        dst2[1] = pt1[1];
        if (pt1[0] > pt2[0]) { dst1[0] = pt1[0] + deltaLen; dst2[0] = pt2[0] - deltaLen; }
        else                 { dst1[0] = pt1[0] - deltaLen; dst2[0] = pt2[0] + deltaLen; }
    } else {
        float k = (float)(pt1[1] - pt2[1]) / (float)(pt1[0] - pt2[0]);
        float ay = orc_abs_ov(pt1[1] - pt2[1]), ax = orc_abs_ov(pt1[0] - pt2[0]); /* :336 (float)abs(...) */
        float theta, zoomY, zoomX;
        if (g_overloads & 2) { /* the double functions: float theta = atan2(double, double); sin(theta) * deltaLen in double */
            theta = (float)m_atan2((double)ay, (double)ax);
            zoomY = (float)(m_sin((double)theta) * (double)deltaLen);
            zoomX = (float)(m_cos((double)theta) * (double)deltaLen);
        } else {
            theta = m_atan2f(ay, ax);
            zoomY = m_sinf(theta) * deltaLen;
            zoomX = m_cosf(theta) * deltaLen;
        }
        if (k > 0) {
            if (pt1[0] > pt2[0]) {
                dst1[0] = pt1[0] + zoomX; dst1[1] = pt1[1] + zoomY;
                dst2[0] = pt2[0] - zoomX; dst2[1] = pt2[1] - zoomY;
            } else {
                dst1[0] = pt1[0] - zoomX; dst1[1] = pt1[1] - zoomY;
                dst2[0] = pt2[0] + zoomX; dst2[1] = pt2[1] + zoomY;
            }
        } else {
            if (pt1[0] < pt2[0]) {
                dst1[0] = pt1[0] - zoomX; dst1[1] = pt1[1] + zoomY;
                dst2[0] = pt2[0] + zoomX; dst2[1] = pt2[1] - zoomY;
            } else {
                dst1[0] = pt1[0] + zoomX; dst1[1] = pt1[1] - zoomY;
                dst2[0] = pt2[0] - zoomX; dst2[1] = pt2[1] + zoomY;
            }
        }
    }
}

static void line_center(const float a[2], const float b[2], float out[2])
{ /* core.cpp:401-404 */
    out[0] = a[0] / 2 + b[0] / 2;
    out[1] = a[1] / 2 + b[1] / 2;
}

void orc_make_armour(const orc_lightblob* a, const orc_lightblob* b, orc_armour* out)
{
    /* core.cpp:26-30: std::sort of two elements by center.x (strict <, ties keep order) */
    const orc_lightblob *L = a, *R = b;
    if (b->center[0] < a->center[0]) { L = b; R = a; }
    /* core.cpp:32-37 */
    float v[4][2];
    v[0][0] = L->vertices[3][0]; v[0][1] = L->vertices[3][1];
    v[1][0] = L->vertices[2][0]; v[1][1] = L->vertices[2][1];
    v[2][0] = R->vertices[1][0]; v[2][1] = R->vertices[1][1];
    v[3][0] = R->vertices[0][0]; v[3][1] = R->vertices[0][1];
    /* core.cpp:39-44 */
    float distanceL = point_distance(v[0], v[1]);
    float distanceR = point_distance(v[2], v[3]);
    float offsetL = roundf((distanceL / 0.50f - distanceL) / 2);
    float offsetR = roundf((distanceR / 0.50f - distanceR) / 2);
    extend_cord(v[0], v[1], offsetL, out->icon[0], out->icon[1]);
    extend_cord(v[3], v[2], offsetR, out->icon[3], out->icon[2]);
    /* core.cpp:46: [OCV] boundingRect of 4 float points -> Rect(floor(min), floor(max)-floor(min)+1) -> Rect2f */
    float minx = out->icon[0][0], maxx = minx, miny = out->icon[0][1], maxy = miny;
    for (int i = 1; i < 4; i++) {
        if (out->icon[i][0] < minx) minx = out->icon[i][0];
        if (out->icon[i][0] > maxx) maxx = out->icon[i][0];
        if (out->icon[i][1] < miny) miny = out->icon[i][1];
        if (out->icon[i][1] > maxy) maxy = out->icon[i][1];
    }
    int ix = (int)floorf(minx), iy = (int)floorf(miny), ax = (int)floorf(maxx), ay = (int)floorf(maxy);
    out->bbox[0] = (float)ix;
    out->bbox[1] = (float)iy;
    out->bbox[2] = (float)(ax - ix + 1);
    out->bbox[3] = (float)(ay - iy + 1);
    /* core.cpp:48, 382-399: CalcPerspective(vertices, vertices), outRatio = 1 */
    float leftHeight = point_distance(v[0], v[1]);
    float rightHeight = point_distance(v[2], v[3]);
    float maxHeight = leftHeight > rightHeight ? leftHeight : rightHeight; /* fmax */
    float sw = maxHeight * 1.0f, sh = maxHeight;
    float c01[2], c23[2], c[2];
    line_center(v[0], v[1], c01);
    line_center(v[2], v[3], c23);
    line_center(c01, c23, c);
    out->vertices[0][0] = c[0] - sw / 2; out->vertices[0][1] = c[1] - sh / 2;
    out->vertices[1][0] = c[0] - sw / 2; out->vertices[1][1] = c[1] + sh / 2;
    out->vertices[2][0] = c[0] + sw / 2; out->vertices[2][1] = c[1] + sh / 2;
    out->vertices[3][0] = c[0] + sw / 2; out->vertices[3][1] = c[1] - sh / 2;
    out->blob_i = -1;
    out->blob_j = -1;
}

/* ------------------------------------------------- objdetect.cpp:55-87 */
int orc_filter_lightblobs(const orc_point* pts, const int32_t* offs, int n_contours, float tilt_max,
                          float ratio_lo, float ratio_hi, double area_lo, double area_hi, int enemy,
                          orc_lightblob* blobs, int cap_blobs, int32_t* n_blobs, int32_t* blob_src,
                          int32_t* neg_idx, int32_t* n_neg, orc_rrect* ellipses)
{
    int np = 0, nn = 0, rc = 0;
    for (int c = 0; c < n_contours; c++) {
        const orc_point* cp = pts + offs[c];
        int n = offs[c + 1] - offs[c];
        if (n < 6) continue; /* :64 */
        double area = orc_contour_area(cp, n);
        if (!(area >= area_lo && area <= area_hi)) continue; /* range<double>::contains, core.h:40-43 */
        int negative_flag = 0;
        orc_rrect ell;
        orc_fit_ellipse_direct(cp, n, &ell); /* :68 */
        /* :69 minAreaRect result is never read -- omitted (no observable effect) */
        float mx = ell.w > ell.h ? ell.w : ell.h, mn = ell.w < ell.h ? ell.w : ell.h;
        float ratio = mx / mn; /* :71-73 */
        if (!(ratio >= ratio_lo && ratio <= ratio_hi)) negative_flag = 1;
        float angle = ell.angle > 90 ? ell.angle - 90 : ell.angle + 90; /* :78 */
        if (orc_abs_ov(angle - 90) > tilt_max) negative_flag = 1;       /* :79 */
        if (negative_flag) {
            if (neg_idx) neg_idx[nn] = c;
            nn++;
        } else {
            if (np < cap_blobs && blobs) {
                orc_make_lightblob(&ell, enemy, &blobs[np]);
                if (blob_src) blob_src[np] = c;
                if (ellipses) ellipses[np] = ell;
            } else rc = -2;
            np++;
        }
    }
    if (n_blobs) *n_blobs = np;
    if (n_neg) *n_neg = nn;
    return rc;
}

/* ------------------------------------------------- objdetect.cpp:114-166 */
int orc_filter_armours(const orc_lightblob* lb, int n, float angle_diff_max, float shear_max,
                       float length_ratio_max, int enemy, orc_armour* out, int cap, int32_t* n_out)
{
    int na = 0, rc = 0;
    if (n < 2) { if (n_out) *n_out = 0; return 0; } /* :120 */
    for (int i = 0; i < n - 1; i++) {
        if (lb[i].target != enemy) continue;
        for (int j = i + 1; j < n; j++) {
            if (lb[j].target != enemy) continue;
            float angle_difference = orc_abs_ov(lb[i].angle - lb[j].angle); /* :131 */
            if (angle_difference > angle_diff_max) continue;
            float y = orc_abs_ov(lb[i].center[1] - lb[j].center[1]);
            float x = orc_abs_ov(lb[i].center[0] - lb[j].center[0]);
            float rect_angle = ov_atan2_deg(y, x); /* :137 */
            float shear_i = orc_abs_ov(lb[i].angle > 90 ? orc_abs_ov(lb[i].angle - rect_angle) - 90
                                                        : orc_abs_ov(180 - lb[i].angle - rect_angle) - 90);
            float shear_j = orc_abs_ov(lb[j].angle > 90 ? orc_abs_ov(lb[j].angle - rect_angle) - 90
                                                        : orc_abs_ov(180 - lb[j].angle - rect_angle) - 90);
            if (shear_i > shear_max || shear_j > shear_max) continue; /* :144 */
            float height_i = lb[i].size[1], height_j = lb[j].size[1];
            float mn = height_i < height_j ? height_i : height_j, mx = height_i < height_j ? height_j : height_i;
            float ratio = mn / mx;
            if (ratio < length_ratio_max) continue; /* :149 */
            if (orc_abs_ov(lb[i].center[1] - lb[j].center[1]) > (lb[i].size[1] + lb[j].size[1]) / 2) continue; /* :153 */
            if (orc_abs_ov(lb[i].center[0] - lb[j].center[0]) > (lb[i].size[1] + lb[j].size[1]) * 2) continue; /* :157 */
            if (na < cap && out) {
                orc_make_armour(&lb[i], &lb[j], &out[na]);
                out[na].blob_i = i;
                out[na].blob_j = j;
            } else rc = -2;
            na++;
        }
    }
    if (n_out) *n_out = na;
    return rc;
}

/* ------------------------------------------------- main.cpp:172-176 */
int orc_detect_frame(const uint8_t* bgr, int w, int h, int stride, const orc_params* p, uint8_t* binary,
                     orc_point* pts, int cap_pts, int32_t* offs, int cap_contours, int32_t* n_contours,
                     orc_lightblob* blobs, int cap_blobs, int32_t* n_blobs, orc_armour* armours, int cap_armours,
                     int32_t* n_armours)
{
    int rc;
    uint8_t* bin = binary ? binary : (uint8_t*)malloc((size_t)w * h);
    orc_point* lp = pts ? pts : (orc_point*)malloc(sizeof(orc_point) * (size_t)cap_pts);
    int32_t* lo = offs ? offs : (int32_t*)malloc(sizeof(int32_t) * ((size_t)cap_contours + 1));
    orc_lightblob* lbp = blobs ? blobs : (orc_lightblob*)malloc(sizeof(orc_lightblob) * (size_t)cap_blobs);
    int32_t nc = 0, npnt = 0, nb = 0, na = 0, nneg = 0;
    if (!bin || !lp || !lo || !lbp) { rc = -3; goto out; }
    rc = orc_extract_binary(bgr, w, h, stride, p->camp, p->lower_bound, p->morph, bin);
    if (rc) goto out;
    rc = orc_find_contours(bin, w, h, lp, cap_pts, lo, cap_contours, &nc, &npnt);
    if (rc) goto out;
    rc = orc_filter_lightblobs(lp, lo, nc, p->tilt_max, p->ratio_lo, p->ratio_hi, p->area_lo, p->area_hi, p->camp, lbp,
                               cap_blobs, &nb, 0, 0, &nneg, 0);
    if (rc) goto out;
    rc = orc_filter_armours(lbp, nb, p->angle_diff_max, p->shear_max, p->length_ratio_max, p->camp, armours, cap_armours, &na);
out:
    if (n_contours) *n_contours = nc;
    if (n_blobs) *n_blobs = nb;
    if (n_armours) *n_armours = na;
    if (!binary) free(bin);
    if (!pts) free(lp);
    if (!offs) free(lo);
    if (!blobs) free(lbp);
    return rc;
}

/* ================================================================================================
 * SURVEY 8f-1 / BASELINE config 5: icon rectification + SVM digit classifier
 * ================================================================================================ */

/* [OCV] cvRound / saturate_cast<int>(double): round half to even */
static int cv_round(double v) { return (int)lrint(v); }
static int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

/* [OCV] cv::solve(A, b, x, DECOMP_LU) for a 6x6 system in double: LU with partial pivoting (hal::LU64f) */
static int lu_solve6(double A[6][6], double b[6])
{
    const int m = 6;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[j][i]) > fabs(A[k][i])) k = j;
        if (fabs(A[k][i]) < DBL_EPSILON * 100) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) { double t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[i][i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[j][i] * d;
            for (int kk = i + 1; kk < m; kk++) A[j][kk] += alpha * A[i][kk];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) s -= A[i][k] * b[k];
        b[i] = s / A[i][i];
    }
    return 1;
}

/* [OCV] cv::getAffineTransform(src[3], dst[3]) -> 2x3 double */
static int get_affine_transform(const float src[3][2], const float dst[3][2], double M[6])
{
    double a[6][6], b[6];
    memset(a, 0, sizeof(a));
    for (int i = 0; i < 3; i++) {
        a[2 * i][0] = a[2 * i + 1][3] = src[i][0];
        a[2 * i][1] = a[2 * i + 1][4] = src[i][1];
        a[2 * i][2] = a[2 * i + 1][5] = 1;
        b[2 * i] = dst[i][0];
        b[2 * i + 1] = dst[i][1];
    }
    if (!lu_solve6(a, b)) { memset(M, 0, 6 * sizeof(double)); return 0; }
    memcpy(M, b, 6 * sizeof(double));
    return 1;
}

/* [OCV] cv::warpAffine(src ROI, dst, M, dsize = src size, INTER_LINEAR, BORDER_CONSTANT 0) for CV_8UC3:
 * the matrix is inverted, source coordinates are computed in 1/1024 px fixed point and quantised to 1/32 px
 * (INTER_BITS = 5), the four taps are blended with 15-bit weights (32 - fx)(32 - fy)*32 ... and rounded. */
static void warp_affine_roi(const uint8_t* src, int sw, int sh, int sstride, const double Min[6], uint8_t* dst /* sw*sh*3 */)
{
    double M[6];
    memcpy(M, Min, sizeof(M));
    {
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        double b1 = -M[0] * M[2] - M[1] * M[5];
        double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    const int AB_BITS = 10, AB_SCALE = 1 << 10, INTER_BITS = 5, TAB = 32;
    const int round_delta = AB_SCALE / TAB / 2;
    for (int y = 0; y < sh; y++) {
        const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < sw; x++) {
            const int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
            const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
            int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
            if (sx > 32767) sx = 32767; if (sx < -32768) sx = -32768; /* saturate_cast<short> */
            if (sy > 32767) sy = 32767; if (sy < -32768) sy = -32768;
            const int fx = X & (TAB - 1), fy = Y & (TAB - 1);
            int w00 = (TAB - fx) * (TAB - fy) * 32, w01 = fx * (TAB - fy) * 32, w10 = (TAB - fx) * fy * 32, w11 = fx * fy * 32;
            if (w00 > 32767) { w00 = 32767; w11 += 1; } /* saturate_cast<short>(32768) and the table's sum correction */
            uint8_t* d = dst + ((size_t)y * sw + x) * 3;
            for (int c = 0; c < 3; c++) {
                int v00 = 0, v01 = 0, v10 = 0, v11 = 0; /* BORDER_CONSTANT, value 0, outside the ROI */
                if (sy >= 0 && sy < sh) {
                    if (sx >= 0 && sx < sw) v00 = src[(size_t)sy * sstride + 3 * sx + c];
                    if (sx + 1 >= 0 && sx + 1 < sw) v01 = src[(size_t)sy * sstride + 3 * (sx + 1) + c];
                }
                if (sy + 1 >= 0 && sy + 1 < sh) {
                    if (sx >= 0 && sx < sw) v10 = src[(size_t)(sy + 1) * sstride + 3 * sx + c];
                    if (sx + 1 >= 0 && sx + 1 < sw) v11 = src[(size_t)(sy + 1) * sstride + 3 * (sx + 1) + c];
                }
                int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
                d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

/* [OCV] cv::resize(src, dst, 20x20, INTER_LINEAR) for CV_8UC3: 11-bit fixed-point coefficients; an exact 2:1
 * decimation is done as INTER_AREA (2x2 box, rounded) */
static void resize_linear_20(const uint8_t* src, int sw, int sh, uint8_t* dst /* 20*20*3 */)
{
    const int dw = ORC_ICON_SIDE, dh = ORC_ICON_SIDE;
    if (sw == 2 * dw && sh == 2 * dh) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < 3; c++) {
                    const uint8_t* s0 = src + ((size_t)(2 * y) * sw + 2 * x) * 3 + c;
                    const uint8_t* s1 = s0 + (size_t)sw * 3;
                    dst[(y * dw + x) * 3 + c] = (uint8_t)((s0[0] + s0[3] + s1[0] + s1[3] + 2) >> 2);
                }
        return;
    }
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int xofs[ORC_ICON_SIDE], yofs[ORC_ICON_SIDE];
    short ia[ORC_ICON_SIDE][2], ib[ORC_ICON_SIDE][2];
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ia[dx][0] = (short)cv_round((1.f - fx) * 2048);
        ia[dx][1] = (short)cv_round(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        ib[dy][0] = (short)cv_round((1.f - fy) * 2048);
        ib[dy][1] = (short)cv_round(fy * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        if (sy0 < 0) sy0 = 0; if (sy0 > sh - 1) sy0 = sh - 1; /* clip(sy + k, 0, ssize.height) */
        if (sy1 < 0) sy1 = 0; if (sy1 > sh - 1) sy1 = sh - 1;
        for (int dx = 0; dx < dw; dx++)
            for (int c = 0; c < 3; c++) {
                int r0, r1;
                const int sx = xofs[dx];
                if (dx < xmax) {
                    r0 = src[((size_t)sy0 * sw + sx) * 3 + c] * ia[dx][0] + src[((size_t)sy0 * sw + sx + 1) * 3 + c] * ia[dx][1];
                    r1 = src[((size_t)sy1 * sw + sx) * 3 + c] * ia[dx][0] + src[((size_t)sy1 * sw + sx + 1) * 3 + c] * ia[dx][1];
                } else {
                    r0 = src[((size_t)sy0 * sw + sx) * 3 + c] * 2048;
                    r1 = src[((size_t)sy1 * sw + sx) * 3 + c] * 2048;
                }
                const int v = (((ib[dy][0] * (r0 >> 4)) >> 16) + ((ib[dy][1] * (r1 >> 4)) >> 16) + 2) >> 2;
                dst[(dy * dw + dx) * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
}

int orc_affine_correction(const uint8_t* bgr, int w, int h, int stride, float icon[4][2], uint8_t* out)
{
    /* imgproc.cpp:11-15 */
    for (int i = 0; i < 4; i++) {
        float vx = icon[i][0] < (float)w - 1 ? icon[i][0] : (float)w - 1; /* std::min(v, cols - 1) */
        icon[i][0] = 0.0f > vx ? 0.0f : vx;                               /* std::max(0, .) */
        float vy = icon[i][1] < (float)h - 1 ? icon[i][1] : (float)h - 1;
        icon[i][1] = 0.0f > vy ? 0.0f : vy;
    }
    /* :17 boundingRect(vector<Point>(vertices)): Point2f -> Point rounds (cvRound); int points: inclusive box */
    int minx = 0, maxx = 0, miny = 0, maxy = 0;
    for (int i = 0; i < 4; i++) {
        const int px = cv_round(icon[i][0]), py = cv_round(icon[i][1]);
        if (i == 0 || px < minx) minx = px;
        if (i == 0 || px > maxx) maxx = px;
        if (i == 0 || py < miny) miny = py;
        if (i == 0 || py > maxy) maxy = py;
    }
    const int bx = minx, by = miny, bw = maxx - minx + 1, bh = maxy - miny + 1;
    memset(out, 0, ORC_SVM_FEATURES);
    if (bw <= 0 || bh <= 0 || bx < 0 || by < 0 || bx + bw > w || by + bh > h) return 1;
    /* :18-27 */
    const float srcPts[3][2] = {{icon[1][0] - (float)bx, icon[1][1] - (float)by},
                                {icon[2][0] - (float)bx, icon[2][1] - (float)by},
                                {icon[0][0] - (float)bx, icon[0][1] - (float)by}};
    const float dstPts[3][2] = {{0, 0}, {(float)bw, 0}, {0, (float)bh}};
    double M[6];
    get_affine_transform(srcPts, dstPts, M); /* :28 */
    /* :30-32 warpAffine(source(box), calibration, warp, Size()) -> same size as the ROI; resize to 20x20 */
    uint8_t* tmp = (uint8_t*)malloc((size_t)bw * bh * 3);
    if (!tmp) return 1;
    warp_affine_roi(bgr + (size_t)by * stride + 3 * (size_t)bx, bw, bh, stride, M, tmp);
    resize_linear_20(tmp, bw, bh, out);
    free(tmp);
    return 0;
}

int orc_svm_predict(const float* x, int n_feat, const float* weights, const double* rho, const int32_t* labels, int n_class)
{
    /* [OCV] SVMImpl::predict for C_SVC, LINEAR kernel after optimize_linear_svm: every decision function is one
     * weight vector; kernel value = float dot accumulated four products at a time into a double */
    int vote[16] = {0};
    int dfi = 0;
    for (int i = 0; i < n_class; i++)
        for (int j = i + 1; j < n_class; j++, dfi++) {
            const float* wv = weights + (size_t)dfi * n_feat;
            double s = 0;
            int k = 0;
            for (; k <= n_feat - 4; k += 4) s += wv[k] * x[k] + wv[k + 1] * x[k + 1] + wv[k + 2] * x[k + 2] + wv[k + 3] * x[k + 3];
            for (; k < n_feat; k++) s += wv[k] * x[k];
            const float kval = (float)(s * 1.0 + 0.0);
            const double sum = -rho[dfi] + 1.0 * kval;
            vote[sum > 0 ? i : j]++;
        }
    int best = 0;
    for (int i = 1; i < n_class; i++)
        if (vote[i] > vote[best]) best = i;
    return labels[best];
}

void orc_classify_armours(const uint8_t* bgr, int w, int h, int stride, orc_armour* armours, int n, const float* weights,
                          const double* rho, const int32_t* labels, int n_class, int32_t* identity, uint8_t* icons)
{
    uint8_t icon[ORC_SVM_FEATURES];
    float feat[ORC_SVM_FEATURES];
    for (int i = 0; i < n; i++) {
        orc_affine_correction(bgr, w, h, stride, armours[i].icon, icon); /* main.cpp:180 */
        for (int k = 0; k < ORC_SVM_FEATURES; k++) feat[k] = (float)icon[k]; /* flatten_image: reshape(1,1), CV_32FC1 */
        identity[i] = orc_svm_predict(feat, ORC_SVM_FEATURES, weights, rho, labels, n_class); /* main.cpp:181 */
        if (icons) memcpy(icons + (size_t)i * ORC_SVM_FEATURES, icon, ORC_SVM_FEATURES);
    }
}
