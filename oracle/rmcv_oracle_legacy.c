/*
 * rmcv_oracle_legacy.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), SURVEY 8f-2:
 * the legacy per-contour matcher
 *   rm::MatchLightBlob     /root/reference/src/objdetect.cpp:9-28
 *   rm::FindLightBlobs     /root/reference/src/objdetect.cpp:30-53
 *   rm::LightBlobOverlap   /root/reference/src/objdetect.cpp:89-112
 * and the OpenCV primitives only this path needs: cv::minAreaRect (convexHull + rotating
 * calipers), cv::boundingRect on int points, cv::mean over a 3-channel u8 ROI.
 *
 * [OCV] marks OpenCV behaviour (un-vendored dependency, >= 4.8.0, vcpkg.json:28-35) restated from
 * the published algorithm as recalled: Sklansky's scan over the points sorted by (x, y, address),
 * four quarter chains, the output assembled counter-clockwise (minAreaRect asks convexHull for
 * clockwise=false since 4.5.1), a cyclic shift that makes the hull's point indices monotone when
 * possible, then Toussaint's rotating calipers in float.  PARITY UNPINNED against real OpenCV, as
 * for the rest of the oracle (rmcv_oracle.h).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../rmcv_amd/csrc/pinned_math.h"
#include "rmcv_oracle.h"

#define ORC_PI 3.1415926535897932384626433832795

static double l_atan2(double y, double x) { return orc_get_math_mode() ? atan2(y, x) : pm_atan2(y, x); }

/* ------------------------------------------------- [OCV] convexHull (int points) */
typedef struct { int32_t x, y, idx; } hpt;

static int hpt_cmp(const void* a, const void* b)
{
    const hpt* p = (const hpt*)a;
    const hpt* q = (const hpt*)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx); /* OpenCV breaks ties by element address = index */
}

static int sgn64(int64_t v) { return (v > 0) - (v < 0); }

/* [OCV] Sklansky_: scan array[start..end] (either direction), keep the chain whose turns have sign sign2 and whose y
 * steps never have sign nsign.  stack receives indices into the sorted array; returns the chain length. */
static int sklansky(const hpt* a, int start, int end, int* stack, int nsign, int sign2)
{
    int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || (a[start].x == a[end].x && a[start].y == a[end].y)) {
        stack[0] = start;
        return 1;
    }
    stack[0] = pprev;
    stack[1] = pcur;
    stack[2] = pnext;
    end += incr;
    while (pnext != end) {
        int cury = a[pcur].y, nexty = a[pnext].y;
        int by = nexty - cury;
        if (sgn64(by) != nsign) {
            int ax = a[pcur].x - a[pprev].x;
            int bx = a[pnext].x - a[pcur].x;
            int ay = cury - a[pprev].y;
            int64_t convexity = (int64_t)ay * bx - (int64_t)ax * by;
            if (sgn64(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur;
                pcur = pnext;
                pnext += incr;
                stack[stacksize] = pnext;
                stacksize++;
            } else {
                if (pprev == start) {
                    pcur = pnext;
                    stack[1] = pcur;
                    pnext += incr;
                    stack[2] = pnext;
                } else {
                    stack[stacksize - 2] = pnext;
                    pcur = pprev;
                    pprev = stack[stacksize - 4];
                    stacksize--;
                }
            }
        } else {
            pnext += incr;
            stack[stacksize - 1] = pnext;
        }
    }
    return --stacksize;
}

/* the body of convexHull once the points are sorted: a[0..total) sorted, hullbuf receives ORIGINAL indices.
 * stack needs total + 2 entries.  clockwise = false (what minAreaRect asks for). */
static int hull_from_sorted(const hpt* a, int total, int* stack, int* hullbuf)
{
    int nout = 0, i;
    int miny_ind = 0, maxy_ind = 0;
    for (i = 1; i < total; i++) {
        int y = a[i].y;
        if (a[miny_ind].y > y) miny_ind = i;
        if (a[maxy_ind].y < y) maxy_ind = i;
    }
    if (a[0].x == a[total - 1].x && a[0].y == a[total - 1].y) {
        hullbuf[nout++] = a[0].idx;
        return nout;
    }
    /* upper half */
    int* tl_stack = stack;
    int tl_count = sklansky(a, 0, maxy_ind, tl_stack, -1, 1);
    int* tr_stack = stack + tl_count;
    int tr_count = sklansky(a, total - 1, maxy_ind, tr_stack, -1, -1);
    { /* !clockwise: swap */
        int* t = tl_stack; tl_stack = tr_stack; tr_stack = t;
        int c = tl_count; tl_count = tr_count; tr_count = c;
    }
    for (i = 0; i < tl_count - 1; i++) hullbuf[nout++] = a[tl_stack[i]].idx;
    for (i = tr_count - 1; i > 0; i--) hullbuf[nout++] = a[tr_stack[i]].idx;
    int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
    /* lower half (clockwise == false: no swap) */
    int* bl_stack = stack;
    int bl_count = sklansky(a, 0, miny_ind, bl_stack, 1, -1);
    int* br_stack = stack + bl_count;
    int br_count = sklansky(a, total - 1, miny_ind, br_stack, 1, 1);
    if (stop_idx >= 0) {
        int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
        if (check_idx == stop_idx || (check_idx >= 0 && a[check_idx].x == a[stop_idx].x && a[check_idx].y == a[stop_idx].y)) {
            /* all points on one line: the bottom part mirrors the top part */
            bl_count = bl_count < 2 ? bl_count : 2;
            br_count = br_count < 2 ? br_count : 2;
        }
    }
    for (i = 0; i < bl_count - 1; i++) hullbuf[nout++] = a[bl_stack[i]].idx;
    for (i = br_count - 1; i > 0; i--) hullbuf[nout++] = a[br_stack[i]].idx;
    /* cyclic shift that makes the index sequence ascending or descending, when one exists */
    if (nout >= 3) {
        int min_idx = 0, max_idx = 0, lt = 0;
        for (i = 1; i < nout; i++) {
            int idx = hullbuf[i];
            lt += hullbuf[i - 1] < idx;
            if (lt > 1 && lt <= i - 2) break;
            if (idx < hullbuf[min_idx]) min_idx = i;
            if (idx > hullbuf[max_idx]) max_idx = i;
        }
        int mmdist = abs(max_idx - min_idx);
        if ((mmdist == 1 || mmdist == nout - 1) && (lt <= 1 || lt >= nout - 2)) {
            int ascending = (max_idx + 1) % nout == min_idx;
            int i0 = ascending ? min_idx : max_idx, j = i0;
            if (i0 > 0) {
                for (i = 0; i < nout; i++) {
                    int curr_idx = stack[i] = hullbuf[j];
                    int next_j = j + 1 < nout ? j + 1 : 0;
                    int next_idx = hullbuf[next_j];
                    if (i < nout - 1 && (ascending != (curr_idx < next_idx))) break;
                    j = next_j;
                }
                if (i == nout) memcpy(hullbuf, stack, (size_t)nout * sizeof(hullbuf[0]));
            }
        }
    }
    return nout;
}

/* cv::convexHull(points, hull, clockwise=false, returnPoints=false): hull_idx gets indices into pts; returns the count */
int orc_convex_hull(const orc_point* pts, int n, int32_t* hull_idx)
{
    if (n <= 0) return 0;
    hpt* a = (hpt*)malloc(sizeof(hpt) * (size_t)n);
    int* stack = (int*)malloc(sizeof(int) * ((size_t)n + 2));
    int* hb = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) { a[i].x = pts[i].x; a[i].y = pts[i].y; a[i].idx = i; }
    qsort(a, (size_t)n, sizeof(hpt), hpt_cmp); /* the comparison is a total order: any sort gives the same array */
    int nout = hull_from_sorted(a, n, stack, hb);
    for (int i = 0; i < nout; i++) hull_idx[i] = hb[i];
    free(a); free(stack); free(hb);
    return nout;
}

/* TEST CROSS-CHECK of the pruning the HIP kernel relies on (rmcv_amd/csrc/device_hull.h): the same scan over four entries
 * per column of the contour's bounding box -- (ymin, lowest index), (ymin, highest index), (ymax, lowest index), (ymax,
 * highest index) -- instead of over all sorted points.  A closed 8-connected border visits every column of its box. */
int orc_convex_hull_pruned(const orc_point* pts, int n, int32_t* hull_idx)
{
    if (n <= 0) return 0;
    int minx = pts[0].x, maxx = pts[0].x;
    for (int i = 1; i < n; i++) { if (pts[i].x < minx) minx = pts[i].x; if (pts[i].x > maxx) maxx = pts[i].x; }
    int W = maxx - minx + 1, total = 4 * W;
    hpt* a = (hpt*)malloc(sizeof(hpt) * (size_t)total);
    int* stack = (int*)malloc(sizeof(int) * ((size_t)total + 2));
    int* hb = (int*)malloc(sizeof(int) * (size_t)total);
    for (int k = 0; k < total; k++) { a[k].x = minx + (k >> 2); a[k].y = 0; a[k].idx = -1; }
    for (int i = 0; i < n; i++) {
        hpt* c = a + 4 * (pts[i].x - minx);
        int y = pts[i].y;
        if (c[0].idx < 0) { for (int s = 0; s < 4; s++) { c[s].y = y; c[s].idx = i; } continue; }
        if (y < c[0].y) { c[0].y = c[1].y = y; c[0].idx = c[1].idx = i; }
        else if (y == c[0].y) c[1].idx = i; /* points arrive in index order: lowest stays in slot 0, highest ends in slot 1 */
        if (y > c[2].y) { c[2].y = c[3].y = y; c[2].idx = c[3].idx = i; }
        else if (y == c[2].y) c[3].idx = i;
    }
    int rc = -1;
    for (int k = 0; k < total; k++) if (a[k].idx < 0) goto out; /* a column without a point: not a closed border */
    rc = hull_from_sorted(a, total, stack, hb);
    for (int i = 0; i < rc; i++) hull_idx[i] = hb[i];
out:
    free(a); free(stack); free(hb);
    return rc;
}

/* ------------------------------------------------- [OCV] rotatingCalipers(CALIPERS_MINAREARECT) + minAreaRect */
static void rotating_calipers(const float* px, const float* py, int n, float out[6])
{
    float minarea = FLT_MAX;
    int i, k;
    float* inv_vect_length = (float*)malloc(sizeof(float) * (size_t)n * 3);
    float* vx = inv_vect_length + n;
    float* vy = vx + n;
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4] = {-1, -1, -1, -1};
    float orientation = 0, base_a, base_b = 0;
    float left_x, right_x, top_y, bottom_y;
    float pt0x = px[0], pt0y = py[0];
    /* kept solution */
    int buf_left = 0, buf_bottom = 0;
    float buf_a = 0, buf_b = 0, buf_w = 0, buf_h = 0;
    left_x = right_x = pt0x;
    top_y = bottom_y = pt0y;
    for (i = 0; i < n; i++) {
        double dx, dy;
        if (pt0x < left_x) left_x = pt0x, left = i;
        if (pt0x > right_x) right_x = pt0x, right = i;
        if (pt0y > top_y) top_y = pt0y, top = i;
        if (pt0y < bottom_y) bottom_y = pt0y, bottom = i;
        int nx = i + 1 < n ? i + 1 : 0;
        float ptx = px[nx], pty = py[nx];
        dx = ptx - pt0x;
        dy = pty - pt0y;
        vx[i] = (float)dx;
        vy[i] = (float)dy;
        inv_vect_length[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        pt0x = ptx;
        pt0y = pty;
    }
    { /* hull orientation */
        double ax = vx[n - 1], ay = vy[n - 1];
        for (i = 0; i < n; i++) {
            double bx = vx[i], by = vy[i];
            double convexity = ax * by - ay * bx;
            if (convexity != 0) {
                orientation = (convexity > 0) ? 1.f : (-1.f);
                break;
            }
            ax = bx;
            ay = by;
        }
        /* OpenCV asserts orientation != 0; a hull with >= 3 points always has a turn */
    }
    base_a = orientation;
    seq[0] = bottom;
    seq[1] = right;
    seq[2] = top;
    seq[3] = left;
    for (k = 0; k < n; k++) {
        float dp[4] = {
            +base_a * vx[seq[0]] + base_b * vy[seq[0]],
            -base_b * vx[seq[1]] + base_a * vy[seq[1]],
            -base_a * vx[seq[2]] - base_b * vy[seq[2]],
            +base_b * vx[seq[3]] - base_a * vy[seq[3]],
        };
        float maxcos = dp[0] * inv_vect_length[seq[0]];
        int main_element = 0;
        for (i = 1; i < 4; ++i) {
            float cosalpha = dp[i] * inv_vect_length[seq[i]];
            if (cosalpha > maxcos) {
                main_element = i;
                maxcos = cosalpha;
            }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vx[pindex] * inv_vect_length[pindex];
            float lead_y = vy[pindex] * inv_vect_length[pindex];
            switch (main_element) {
            case 0: base_a = lead_x; base_b = lead_y; break;
            case 1: base_a = lead_y; base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
        {
            float dx = px[seq[1]] - px[seq[3]];
            float dy = py[seq[1]] - py[seq[3]];
            float width = dx * base_a + dy * base_b;
            dx = px[seq[2]] - px[seq[0]];
            dy = py[seq[2]] - py[seq[0]];
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                buf_left = seq[3];
                buf_a = base_a;
                buf_w = width;
                buf_b = base_b;
                buf_h = height;
                buf_bottom = seq[0];
            }
        }
    }
    {
        float A1 = buf_a, B1 = buf_b, A2 = -buf_b, B2 = buf_a;
        float C1 = A1 * px[buf_left] + py[buf_left] * B1;
        float C2 = A2 * px[buf_bottom] + py[buf_bottom] * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        float ox = (C1 * B2 - C2 * B1) * idet;
        float oy = (A1 * C2 - A2 * C1) * idet;
        out[0] = ox;
        out[1] = oy;
        out[2] = A1 * buf_w;
        out[3] = B1 * buf_w;
        out[4] = A2 * buf_h;
        out[5] = B2 * buf_h;
    }
    free(inv_vect_length);
}

/* cv::minAreaRect(contour) */
void orc_min_area_rect(const orc_point* pts, int n, orc_rrect* box)
{
    box->cx = box->cy = box->w = box->h = box->angle = 0;
    if (n <= 0) return;
    int32_t* hi = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int nh = orc_convex_hull(pts, n, hi);
    float* hx = (float*)malloc(sizeof(float) * (size_t)nh * 2);
    float* hy = hx + nh;
    for (int i = 0; i < nh; i++) { hx[i] = (float)pts[hi[i]].x; hy[i] = (float)pts[hi[i]].y; }
    if (nh > 2) {
        float out[6];
        rotating_calipers(hx, hy, nh, out);
        box->cx = out[0] + (out[2] + out[4]) * 0.5f;
        box->cy = out[1] + (out[3] + out[5]) * 0.5f;
        box->w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
        box->h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
        box->angle = (float)l_atan2((double)out[3], (double)out[2]);
    } else if (nh == 2) {
        box->cx = (hx[0] + hx[1]) * 0.5f;
        box->cy = (hy[0] + hy[1]) * 0.5f;
        double dx = hx[1] - hx[0], dy = hy[1] - hy[0];
        box->w = (float)sqrt(dx * dx + dy * dy);
        box->h = 0;
        box->angle = (float)l_atan2(dy, dx);
    } else if (nh == 1) {
        box->cx = hx[0];
        box->cy = hy[0];
    }
    box->angle = (float)(box->angle * 180 / ORC_PI);
    free(hi); free(hx);
}

/* ------------------------------------------------- objdetect.cpp:9-28 */
int orc_match_lightblob(const orc_point* pts, int n, float min_ratio, float max_ratio, float tilt_angle, float min_area,
                        float max_area, int fit_ellipse, orc_rrect* box_out)
{
    if (n < 6) return 0; /* :12 */
    double area = orc_contour_area(pts, n);
    if (area < min_area || area > max_area) return 0;
    orc_rrect ellipse, box;
    orc_fit_ellipse_direct(pts, n, &ellipse); /* :15 */
    if (fit_ellipse) box = ellipse;
    else orc_min_area_rect(pts, n, &box);     /* :16 */
    float mx = box.w > box.h ? box.w : box.h, mn = box.w < box.h ? box.w : box.h;
    float ratio = mx / mn;                    /* :19 */
    if (ratio > max_ratio || ratio < min_ratio) return 0;
    float angle = ellipse.angle > 90 ? ellipse.angle - 90 : ellipse.angle + 90; /* :23 */
    if (orc_abs_ov(angle - 90) > tilt_angle) return 0; /* :24, an unqualified abs on a float: SURVEY A.6 */
    *box_out = box;
    return 1;
}

/* [OCV] boundingRect of int points: inclusive box */
void orc_bounding_rect(const orc_point* pts, int n, int32_t rect[4])
{
    int minx = pts[0].x, maxx = pts[0].x, miny = pts[0].y, maxy = pts[0].y;
    for (int i = 1; i < n; i++) {
        if (pts[i].x < minx) minx = pts[i].x;
        if (pts[i].x > maxx) maxx = pts[i].x;
        if (pts[i].y < miny) miny = pts[i].y;
        if (pts[i].y > maxy) maxy = pts[i].y;
    }
    rect[0] = minx; rect[1] = miny; rect[2] = maxx - minx + 1; rect[3] = maxy - miny + 1;
}

/* objdetect.cpp:43-51: [OCV] cv::mean = per-channel integer sum x (1/N) in double, then the three-way comparison */
int orc_camp_from_mean(const uint8_t* bgr, int stride, const int32_t rect[4])
{
    uint64_t s[3] = {0, 0, 0};
    for (int y = rect[1]; y < rect[1] + rect[3]; y++) {
        const uint8_t* row = bgr + (size_t)y * stride + (size_t)rect[0] * 3;
        for (int x = 0; x < rect[2]; x++) { s[0] += row[3 * x]; s[1] += row[3 * x + 1]; s[2] += row[3 * x + 2]; }
    }
    double inv = 1. / ((double)rect[2] * rect[3]);
    double m0 = (double)s[0] * inv, m1 = (double)s[1] * inv, m2 = (double)s[2] * inv;
    if (m1 > m0 && m1 > m2) return ORC_CAMP_GUIDELIGHT;
    return m0 > m2 ? ORC_CAMP_BLUE : ORC_CAMP_RED;
}

/* objdetect.cpp:30-53 */
int orc_find_lightblobs(const uint8_t* bgr, int w, int h, int stride, const orc_point* pts, const int32_t* offs,
                        int n_contours, float min_ratio, float max_ratio, float tilt_angle, float min_area, float max_area,
                        int fit_ellipse, orc_lightblob* blobs, int cap_blobs, int32_t* n_blobs, int32_t* blob_src,
                        orc_rrect* boxes)
{
    int nb = 0, rc = 0;
    (void)w; (void)h;
    for (int c = 0; c < n_contours; c++) {
        const orc_point* cp = pts + offs[c];
        int n = offs[c + 1] - offs[c];
        orc_rrect box;
        if (!orc_match_lightblob(cp, n, min_ratio, max_ratio, tilt_angle, min_area, max_area, fit_ellipse, &box)) continue;
        int32_t rect[4];
        orc_bounding_rect(cp, n, rect);
        int camp = orc_camp_from_mean(bgr, stride, rect);
        if (nb < cap_blobs && blobs) {
            orc_make_lightblob(&box, camp, &blobs[nb]);
            if (blob_src) blob_src[nb] = c;
            if (boxes) boxes[nb] = box;
        } else rc = -2;
        nb++;
    }
    if (n_blobs) *n_blobs = nb;
    return rc;
}

/* objdetect.cpp:89-112.  The reference's bound check lets rightIndex == size() through and then reads one past the end
 * (SURVEY Appendix B); undefined behaviour cannot be mirrored, so that case returns -1 here and in the product. */
int orc_lightblob_overlap(const orc_lightblob* lb, int n, int left, int right)
{
    if (left < 0 || right > n || right - left < 2) return 0;
    if (right == n) return -1;
    if (lb[left].target != lb[right].target) return 0;
    float a = lb[left].vertices[1][1] < lb[left].vertices[2][1] ? lb[left].vertices[1][1] : lb[left].vertices[2][1];
    float b = lb[right].vertices[1][1] < lb[right].vertices[2][1] ? lb[right].vertices[1][1] : lb[right].vertices[2][1];
    float lower_y = a < b ? a : b;
    a = lb[left].vertices[0][1] > lb[left].vertices[3][1] ? lb[left].vertices[0][1] : lb[left].vertices[3][1];
    b = lb[right].vertices[0][1] > lb[right].vertices[3][1] ? lb[right].vertices[0][1] : lb[right].vertices[3][1];
    float upper_y = a > b ? a : b;
    for (int i = left; i < right; i++) {
        if (lb[i].target != lb[left].target) continue;
        if (lb[i].center[0] > lb[left].center[0] && lb[i].center[0] < lb[right].center[0] && lb[i].center[1] > lower_y &&
            lb[i].center[1] < upper_y)
            return 1;
    }
    return 0;
}
