/*
 * rmcv_oracle_track.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), SURVEY 8f-4: what is OBSERVABLE of the tracker
 *   rm::armour::max_IoU       /root/reference/src/core.cpp:144-162   ([OCV] cv::Rect2f operator&, area())
 *   rm::armour::identity_max  /root/reference/src/core.cpp:124-142   (soft-max over the identity histogram)
 * The Kalman state of rm::armour::reset/update (core.cpp:51-122) is private and never read anywhere in the reference
 * (executable/main.cpp:57-88 stops at "decide which armour to shoot"), so it has no observable result to restate.
 * [OCV] Rect_<float>::operator& is the overflow-safe form of OpenCV >= 4.5 as recalled; parity unpinned (rmcv_oracle.h).
 */
#include <math.h>

#include "rmcv_oracle.h"

typedef struct { float x, y, w, h; } rectf;

static int rect_empty(const rectf* r) { return r->w <= 0 || r->h <= 0; }

static rectf rect_and(rectf a, rectf b)
{
    const rectf zero = {0, 0, 0, 0};
    if (rect_empty(&a) || rect_empty(&b)) return zero;
    const rectf* rx_min = (a.x < b.x) ? &a : &b;
    const rectf* rx_max = (a.x < b.x) ? &b : &a;
    const rectf* ry_min = (a.y < b.y) ? &a : &b;
    const rectf* ry_max = (a.y < b.y) ? &b : &a;
    if ((rx_min->x < 0 && rx_min->x + rx_min->w < rx_max->x) || (ry_min->y < 0 && ry_min->y + ry_min->h < ry_max->y)) return zero;
    rectf o;
    const float w1 = rx_min->w - (rx_max->x - rx_min->x), h1 = ry_min->h - (ry_max->y - ry_min->y);
    if (rx_max->w < w1) o.w = rx_max->w; else o.w = w1;
    if (ry_max->h < h1) o.h = ry_max->h; else o.h = h1;
    o.x = rx_max->x;
    o.y = ry_max->y;
    if (rect_empty(&o)) return zero;
    return o;
}

/* core.cpp:144-162: index of the armour with the largest IoU (> 0, first on ties) and that IoU; index -1 when none overlaps */
void orc_max_iou(const orc_armour* self, const orc_armour* list, int n, int32_t* index, float* iou_out)
{
    int idx = -1;
    float max = 0;
    const rectf me = {self->bbox[0], self->bbox[1], self->bbox[2], self->bbox[3]};
    for (int i = 0; i < n; i++) {
        const rectf other = {list[i].bbox[0], list[i].bbox[1], list[i].bbox[2], list[i].bbox[3]};
        const rectf in = rect_and(me, other);
        const float union_area = me.w * me.h + other.w * other.h - in.w * in.h;
        const float iou = in.w * in.h / union_area;
        if (iou > max) {
            max = iou;
            idx = i;
        }
    }
    *index = idx;
    *iou_out = max;
}

/* core.cpp:124-142: identity_history is a std::map<int,int> (iterated in key order): ids must be ascending */
void orc_identity_max(const int32_t* ids, const int32_t* counts, int n, int32_t* max_id, double* prob_out)
{
    double sum = 0;
    for (int i = 0; i < n; i++) sum += exp((double)counts[i]);
    double max = 0;
    int id = -1;
    for (int i = 0; i < n; i++) {
        const double prob = exp((double)counts[i]) / sum;
        if (prob > max) {
            max = prob;
            id = ids[i];
        }
    }
    *max_id = id;
    *prob_out = max;
}
