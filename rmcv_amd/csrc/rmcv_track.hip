// rmcv_track.hip -- tracker state of rm::armour with everything readable: rm::armour::reset / update(const armour&) /
// update(int64) (/root/reference/src/core.cpp:51-122) and the association loop of the tracking thread
// (executable/main.cpp:57-88).  Host-side by nature: sequential per target, a handful of armours per frame, 6x6 fp64.
//
// The filter is cv::KalmanFilter(6, 6, 0, CV_64F) (core.cpp:21); [OCV] init / predict / correct are restated as recalled from
// OpenCV's video/src/kalman.cpp (gemm = sequential k-sums in double; the gain through cv::solve(..., DECOMP_SVD) = one-sided
// Jacobi SVD + back substitution), parity unpinned like every [OCV] piece (oracle/rmcv_oracle.h).
#include <math.h>
#include <string.h>

#include <vector>

#include <algorithm>

#include "../../include/rmcv_abi.h"

namespace {

typedef double M6[36];

// C = A * B (6x6, row-major); every entry a sequential sum over k ([OCV] GEMMSingleMul)
void mul(const double* A, const double* B, double* C, int n, int m, int p) // (n x m) * (m x p)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += A[i * m + k] * B[k * p + j];
            C[i * p + j] = s;
        }
}
// C = A * B^T + D  ([OCV] gemm(A, B, 1, D, 1, C, GEMM_2_T))
void mul_bt_add(const double* A, const double* B, const double* D, double* C)
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
void jacobi_svd(double* At, double* W, double* Vt, int m, int n)
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
    const int max_iter = std::max(m, 30);
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
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
                changed = true;
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
            std::swap(Wd[i], Wd[j]);
            for (int k = 0; k < m; k++) std::swap(At[i * m + k], At[j * m + k]);
            for (int k = 0; k < n; k++) std::swap(Vt[i * n + k], Vt[j * n + k]);
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
void solve_svd(const double* A, const double* B, double* X)
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

void set_identity(double* M, double v)
{
    for (int i = 0; i < 36; i++) M[i] = 0;
    for (int i = 0; i < 6; i++) M[i * 7] = v;
}

// [OCV] KalmanFilter::predict() without control
void kf_predict(rmcv_track* t)
{
    double temp1[36];
    mul(t->transition, t->state_post, t->state_pre, 6, 6, 1);
    mul(t->transition, t->error_cov_post, temp1, 6, 6, 6);
    mul_bt_add(temp1, t->transition, t->process_noise_cov, t->error_cov_pre);
    memcpy(t->state_post, t->state_pre, sizeof(t->state_pre));
    memcpy(t->error_cov_post, t->error_cov_pre, sizeof(t->error_cov_pre));
}

// [OCV] KalmanFilter::correct(measurement)
void kf_correct(rmcv_track* t)
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

} // namespace

extern "C" {

void rmcv_track_init(rmcv_track* t, const rmcv_armour* a, int32_t identity, int64_t timestamp, const double position[3])
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

void rmcv_track_reset(rmcv_track* t, double process_noise, double measurement_noise, double error)
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

int rmcv_track_update(rmcv_track* t, const rmcv_track* obs, double tick_frequency)
{ // src/core.cpp:74-108: update(const armour& new_observation)
    if (!t || !obs || !(tick_frequency > 0)) return RMCV_ERR_BAD_ARG;
    { // identity_history[new_observation.identity]++ (a std::map: ids stay ascending)
        int k = 0;
        while (k < t->n_ids && t->ids[k] < obs->identity) k++;
        if (k < t->n_ids && t->ids[k] == obs->identity) t->counts[k]++;
        else {
            if (t->n_ids >= RMCV_TRACK_IDS) return RMCV_ERR_CAPACITY;
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
    return RMCV_OK;
}

int rmcv_track_predict(rmcv_track* t, int64_t new_timestamp, double tick_frequency)
{ // src/core.cpp:110-122: update(int64 new_timestamp)
    if (!t || !(tick_frequency > 0)) return RMCV_ERR_BAD_ARG;
    if (!t->initialized) return RMCV_OK;
    const int64_t delta_tick = new_timestamp - t->timestamp;
    const double dt = (double)delta_tick / tick_frequency;
    t->transition[0 * 6 + 3] = dt;
    t->transition[1 * 6 + 4] = dt;
    t->transition[2 * 6 + 5] = dt;
    kf_predict(t);
    return RMCV_OK;
}

int rmcv_track_step(rmcv_track* tracking, int32_t* n_tracking, int cap, rmcv_track* obs, int32_t* n_obs, double tick_frequency)
{ // executable/main.cpp:60-85, one pass of the tracking thread's loop
    if (!tracking || !n_tracking || !n_obs || *n_tracking < 0 || *n_obs < 0 || (*n_obs > 0 && !obs) || !(tick_frequency > 0)) return RMCV_ERR_BAD_ARG;
    int nt = *n_tracking, no = *n_obs;
    if (no == 0) return RMCV_OK; // :61
    // The reference's vectors grow without bound; here `cap` is the caller's.  What the pass will leave behind is counted FIRST, on
    // indices only: the matching depends on the bounding boxes (which update() never touches) and on the lost counts, so a dry run
    // of the loop below tells exactly how many targets survive and how many observations stay unmatched.  Only if those do not fit
    // is RMCV_ERR_CAPACITY returned -- before anything is changed.  (A steady state of N targets matched by N observations needs
    // cap >= N, as the reference's loop does, not 2 N.)
    if (nt + no > cap) {
        std::vector<int> tg((size_t)nt), ob((size_t)no);
        for (int i = 0; i < nt; i++) tg[(size_t)i] = i;
        for (int k = 0; k < no; k++) ob[(size_t)k] = k;
        if (nt > 0)
            for (size_t i = 0; i < tg.size(); i++) {
                int index = -1;
                float iou = 0;
                for (size_t k = 0; k < ob.size(); k++) {
                    int32_t hit = -1;
                    float v = 0;
                    rmcv_max_iou(&tracking[tg[i]].armour, &obs[ob[k]].armour, 1, &hit, &v);
                    if (hit == 0 && v > iou) { iou = v; index = (int)k; }
                }
                if (iou > 0.5f) ob.erase(ob.begin() + index);
                else if (tracking[tg[i]].lost_count > 25) tg.erase(tg.begin() + (long)i); // (and the loop's i++ skips the one that moved in)
            }
        if ((int)(tg.size() + ob.size()) > cap) return RMCV_ERR_CAPACITY;
    }
    if (nt == 0) { // :63-67
        memcpy(tracking, obs, (size_t)no * sizeof(rmcv_track));
        *n_tracking = no;
        return RMCV_OK;
    }
    for (int i = 0; i < nt; i++) { // :69-81
        int32_t index = -1;
        float iou = 0;
        for (int k = 0; k < no; k++) { // armour::max_IoU over the remaining observations (src/core.cpp:144-162), any number of them
            int32_t hit = -1;
            float v = 0;
            rmcv_max_iou(&tracking[i].armour, &obs[k].armour, 1, &hit, &v);
            if (hit == 0 && v > iou) { // `iou > max` with max starting at 0: the first of equal maxima wins, as in the reference's loop
                iou = v;
                index = k;
            }
        }
        if (iou > 0.5f) {
            const int rc = rmcv_track_update(&tracking[i], &obs[index], tick_frequency);
            if (rc) { // only the identity table can overflow (RMCV_TRACK_IDS), and it does before the target is touched: hand back
                *n_tracking = nt; // consistent lists -- what has been processed so far stays processed, nothing is duplicated
                *n_obs = no;
                return rc;
            }
            for (int k = index; k + 1 < no; k++) obs[k] = obs[k + 1]; // armours->erase(begin() + index)
            no--;
        } else if (tracking[i].lost_count++ > 25) {
            for (int k = i; k + 1 < nt; k++) tracking[k] = tracking[k + 1]; // tracking.erase(begin() + i) ...
            nt--; // ... and the loop's i++ then SKIPS the target that moved into slot i (the reference does)
        } else {
            rmcv_track_predict(&tracking[i], tracking[i].timestamp, tick_frequency);
        }
    }
    memcpy(tracking + nt, obs, (size_t)no * sizeof(rmcv_track)); // tracking.insert(end(), armours...)
    *n_tracking = nt + no;
    *n_obs = 0;
    return RMCV_OK;
}

} // extern "C"
