/*
 * rmcv_oracle_track.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), SURVEY 8f-4: what is OBSERVABLE of the tracker
 *   rm::armour::max_IoU       /root/reference/src/core.cpp:144-162   ([OCV] cv::Rect2f operator&, area())
 *   rm::armour::identity_max  /root/reference/src/core.cpp:124-142   (soft-max over the identity histogram)
 *   rm::armour::reset / update(const armour&) / update(int64)   /root/reference/src/core.cpp:51-122
 *   the tracking thread's association loop                        /root/reference/executable/main.cpp:60-85
 * The reference keeps the filter state private; here it is a plain struct so that it can be compared.  [OCV] cv::KalmanFilter
 * (init / predict / correct, gemm as sequential k-sums, cv::solve(DECOMP_SVD) = one-sided Jacobi SVD + back substitution) is
 * restated as recalled from OpenCV's kalman.cpp / lapack.cpp.
 * [OCV] Rect_<float>::operator& is the overflow-safe form of OpenCV >= 4.5 as recalled; parity unpinned (rmcv_oracle.h).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

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

/* ---- tracker state (see the header of this file) ---------------------------------------------------------------------- */

// C = A * B (6x6, row-major); every entry a sequential sum over k ([OCV] GEMMSingleMul)
static void mul(const double* A, const double* B, double* C, int n, int m, int p) // (n x m) * (m x p)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += A[i * m + k] * B[k * p + j];
            C[i * p + j] = s;
        }
}
// C = A * B^T + D  ([OCV] gemm(A, B, 1, D, 1, C, GEMM_2_T))
static void mul_bt_add(const double* A, const double* B, const double* D, double* C)
{
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += A[i * 6 + k] * B[j * 6 + k];
            C[i * 6 + j] = s + D[i * 6 + j];
        }
}

// [OCV] JacobiSVDImpl_<double> on At (n rows of length m: the TRANSPOSE of the m x n matrix), as recalled: one-sided Jacobi
// (Hestenes), eps = DBL_EPSILON * 10, at most max(m, 30) sweeps, singular values sorted descending; At's rows become the
// left singular vectors (scaled to unit length), Vt the right ones.
static void jacobi_svd(double* At, double* W, double* Vt, int m, int n)
{
    const double eps = 2.220446049250313e-16 * 10, minval = 2.2250738585072014e-308;
    double Wd[6];
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
        Wd[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * n + k] = 0;
        Vt[i * n + i] = 1;
    }
    const int max_iter = (m > 30 ? m : 30);
    for (int iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double *Ai = At + i * m, *Aj = At + j * m;
                double a = Wd[i], p = 0, b = Wd[j];
                for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                Wd[i] = a;
                Wd[j] = b;
                changed = 1;
                double *Vi = Vt + i * n, *Vj = Vt + j * n;
                for (int k = 0; k < n; k++) {
                    const double t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0;
                    Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) sd += At[i * m + k] * At[i * m + k];
        Wd[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (Wd[j] < Wd[k]) j = k;
        if (i != j) {
            { double t_ = Wd[i]; Wd[i] = Wd[j]; Wd[j] = t_; }
            for (int k = 0; k < m; k++) { double t_ = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t_; }
            for (int k = 0; k < n; k++) { double t_ = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t_; }
        }
    }
    for (int i = 0; i < n; i++) {
        W[i] = Wd[i];
        // [OCV] a null singular value gets a random unit vector here; a 6x6 innovation covariance H P H^T + R with R > 0 has none
        const double s = Wd[i] > minval ? 1 / Wd[i] : 0.;
        for (int k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

// X = A^-1 * B for 6x6 A, B: [OCV] cv::solve(A, B, X, DECOMP_SVD) = JacobiSVD of A^T's rows + SVBkSb (threshold 2 eps * sum w)
static void solve_svd(const double* A, const double* B, double* X)
{
    double At[36], W[6], Vt[36], buffer[6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) At[i * 6 + j] = A[j * 6 + i];
    jacobi_svd(At, W, Vt, 6, 6); // rows of At: u_i; rows of Vt: v_i
    double threshold = 0;
    for (int i = 0; i < 6; i++) threshold += W[i];
    threshold *= 2.220446049250313e-16 * 2;
    for (int i = 0; i < 36; i++) X[i] = 0;
    for (int i = 0; i < 6; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int j = 0; j < 6; j++) buffer[j] = 0;
        for (int k = 0; k < 6; k++) { // MatrAXPY: buffer += u_i[k] * B[k][:]
            const double s = At[i * 6 + k];
            for (int j = 0; j < 6; j++) buffer[j] += s * B[k * 6 + j];
        }
        for (int j = 0; j < 6; j++) buffer[j] *= wi;
        for (int k = 0; k < 6; k++) { // MatrAXPY: X[k][:] += v_i[k] * buffer
            const double s = Vt[i * 6 + k];
            for (int j = 0; j < 6; j++) X[k * 6 + j] += s * buffer[j];
        }
    }
}

static void set_identity(double* M, double v)
{
    for (int i = 0; i < 36; i++) M[i] = 0;
    for (int i = 0; i < 6; i++) M[i * 7] = v;
}

// [OCV] KalmanFilter::predict() without control
static void kf_predict(orc_track* t)
{
    double temp1[36];
    mul(t->transition, t->state_post, t->state_pre, 6, 6, 1);
    mul(t->transition, t->error_cov_post, temp1, 6, 6, 6);
    mul_bt_add(temp1, t->transition, t->process_noise_cov, t->error_cov_pre);
    memcpy(t->state_post, t->state_pre, sizeof(t->state_pre));
    memcpy(t->error_cov_post, t->error_cov_pre, sizeof(t->error_cov_pre));
}

// [OCV] KalmanFilter::correct(measurement)
static void kf_correct(orc_track* t)
{
    double temp2[36], temp3[36], temp4[36], temp5[6], hx[6], kt2[36];
    mul(t->measurement_matrix, t->error_cov_pre, temp2, 6, 6, 6);
    mul_bt_add(temp2, t->measurement_matrix, t->measurement_noise_cov, temp3);
    solve_svd(temp3, temp2, temp4);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) t->gain[i * 6 + j] = temp4[j * 6 + i];
    mul(t->measurement_matrix, t->state_pre, hx, 6, 6, 1);
    for (int i = 0; i < 6; i++) temp5[i] = t->measurement[i] - hx[i];
    mul(t->gain, temp5, hx, 6, 6, 1);
    for (int i = 0; i < 6; i++) t->state_post[i] = t->state_pre[i] + hx[i];
    mul(t->gain, temp2, kt2, 6, 6, 6);
    for (int i = 0; i < 36; i++) t->error_cov_post[i] = t->error_cov_pre[i] - kt2[i];
}


void orc_track_init(orc_track* t, const orc_armour* a, int32_t identity, int64_t timestamp, const double position[3])
{
    if (!t) return;
    memset(t, 0, sizeof(*t));
    if (a) t->armour = *a;
    t->identity = identity; // executable/main.cpp:181
    t->timestamp = timestamp;
    if (position)
        for (int i = 0; i < 3; i++) t->position[i] = position[i];
    // cv::KalmanFilter::init(6, 6, 0, CV_64F): transition and both noise covariances identity, everything else zero
    set_identity(t->transition, 1.0);
    set_identity(t->process_noise_cov, 1.0);
    set_identity(t->measurement_noise_cov, 1.0);
}

void orc_track_reset(orc_track* t, double process_noise, double measurement_noise, double error)
{ // src/core.cpp:51-72
    if (!t) return;
    set_identity(t->measurement_matrix, 1.0);
    set_identity(t->process_noise_cov, process_noise);
    set_identity(t->measurement_noise_cov, measurement_noise);
    set_identity(t->error_cov_post, error);
    for (int i = 0; i < 6; i++) t->measurement[i] = 0;
    set_identity(t->transition, 1.0);
    t->transition[0 * 6 + 3] = t->transition[1 * 6 + 4] = t->transition[2 * 6 + 5] = 1.0;
    t->initialized = 0;
}

int orc_track_update(orc_track* t, const orc_track* obs, double tick_frequency)
{ // src/core.cpp:74-108: update(const armour& new_observation)
    if (!t || !obs || !(tick_frequency > 0)) return -1;
    { // identity_history[new_observation.identity]++ (a std::map: ids stay ascending)
        int k = 0;
        while (k < t->n_ids && t->ids[k] < obs->identity) k++;
        if (k < t->n_ids && t->ids[k] == obs->identity) t->counts[k]++;
        else {
            if (t->n_ids >= ORC_TRACK_IDS) return -2;
            for (int j = t->n_ids; j > k; j--) { t->ids[j] = t->ids[j - 1]; t->counts[j] = t->counts[j - 1]; }
            t->ids[k] = obs->identity;
            t->counts[k] = 1;
            t->n_ids++;
        }
    }
    if (t->initialized) {
        const int64_t delta_tick = obs->timestamp - t->timestamp;
        const double dt = (double)delta_tick / tick_frequency;
        t->transition[0 * 6 + 3] = dt;
        t->transition[1 * 6 + 4] = dt;
        t->transition[2 * 6 + 5] = dt;
        kf_predict(t);
        t->measurement[3] = (obs->position[0] - t->measurement[0]) / dt;
        t->measurement[4] = (obs->position[1] - t->measurement[1]) / dt;
        t->measurement[5] = (obs->position[2] - t->measurement[2]) / dt;
        t->measurement[0] = obs->position[0];
        t->measurement[1] = obs->position[1];
        t->measurement[2] = obs->position[2];
        kf_correct(t);
    } else {
        t->measurement[0] = obs->position[0];
        t->measurement[1] = obs->position[1];
        t->measurement[2] = obs->position[2];
        kf_correct(t); // no predict yet: errorCovPre is still zero, so this correction leaves the state at zero (as in the reference)
        t->initialized = 1;
    }
    t->timestamp = obs->timestamp;
    return 0;
}

int orc_track_predict(orc_track* t, int64_t new_timestamp, double tick_frequency)
{ // src/core.cpp:110-122: update(int64 new_timestamp)
    if (!t || !(tick_frequency > 0)) return -1;
    if (!t->initialized) return 0;
    const int64_t delta_tick = new_timestamp - t->timestamp;
    const double dt = (double)delta_tick / tick_frequency;
    t->transition[0 * 6 + 3] = dt;
    t->transition[1 * 6 + 4] = dt;
    t->transition[2 * 6 + 5] = dt;
    kf_predict(t);
    return 0;
}

int orc_track_step(orc_track* tracking, int32_t* n_tracking, int cap, orc_track* obs, int32_t* n_obs, double tick_frequency)
{ // executable/main.cpp:60-85, one pass of the tracking thread's loop
    if (!tracking || !n_tracking || !n_obs || *n_tracking < 0 || *n_obs < 0 || (*n_obs > 0 && !obs) || !(tick_frequency > 0)) return -1;
    int nt = *n_tracking, no = *n_obs;
    if (no == 0) return 0; // :61
    if (nt == 0) {               // :63-67
        if (no > cap) return -2;
        memcpy(tracking, obs, (size_t)no * sizeof(orc_track));
        *n_tracking = no;
        return 0;
    }
    for (int i = 0; i < nt; i++) { // :69-81
        int32_t index = -1;
        float iou = 0;
        { // armour::max_IoU over the remaining observations (src/core.cpp:144-162)
            orc_armour* boxes = (orc_armour*)malloc((size_t)no * sizeof(orc_armour) + 1); // std::vector<armour>: no bound
            if (!boxes) return -3;
            for (int k = 0; k < no; k++) boxes[k] = obs[k].armour;
            orc_max_iou(&tracking[i].armour, boxes, no, &index, &iou);
            free(boxes);
        }
        if (iou > 0.5f) {
            int rc = orc_track_update(&tracking[i], &obs[index], tick_frequency);
            if (rc) return rc;
            for (int k = index; k + 1 < no; k++) obs[k] = obs[k + 1]; // armours->erase(begin() + index)
            no--;
        } else if (tracking[i].lost_count++ > 25) {
            for (int k = i; k + 1 < nt; k++) tracking[k] = tracking[k + 1]; // tracking.erase(begin() + i) ...
            nt--; // ... and the loop's i++ then SKIPS the target that moved into slot i (the reference does)
        } else {
            orc_track_predict(&tracking[i], tracking[i].timestamp, tick_frequency);
        }
    }
    if (nt + no > cap) return -2;
    memcpy(tracking + nt, obs, (size_t)no * sizeof(orc_track)); // tracking.insert(end(), armours...)
    *n_tracking = nt + no;
    *n_obs = 0;
    return 0;
}

