// k_pnp.hip -- "next" row SURVEY 8f-3: the pose of every armour.
//   rm::solve_PnP            /root/reference/src/mobility.cpp:166-190    (cv::solvePnP, SOLVEPNP_IPPE_SQUARE)
//   camera -> world position /root/reference/executable/main.cpp:183-192
//
// [OCV] solvePnP(IPPE_SQUARE) = undistortPoints (5 fixed-point iterations, result narrowed to float) + IPPE for a square
// (Collins & Bartoli 2014): homography of the square in closed form, its Jacobian at the origin, the two candidate
// rotations, the least-squares translation of each, the pose with the smaller reprojection error.  A few hundred fp64
// operations per armour with no parallelism inside, a handful of armours per frame: one LANE per armour, one wavefront
// per frame.  Operation order is the CPU restatement's (oracle/rmcv_oracle_pnp.c); sqrt and divide are IEEE, acos / sin
// come from pinned_math.h, no FMA contraction -- the results are bit-identical to it.
#include <float.h>

#include "device_fit.h"
#include "rmcv_internal.h"

namespace rmcv {

__device__ inline void undistort_point(float u, float v, const double* K, const double* k, float* ox, float* oy)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) { // TermCriteria(MAX_ITER, 5, 0.01): exactly five rounds
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {
            x = ((double)u - cx) * ifx;
            y = ((double)v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *ox = (float)x;
    *oy = (float)y;
}

// canonical square (-h, h), (h, h), (h, -h), (-h, -h) -> image points q0..q3, H[8] = 1
__device__ inline void homography_from_square(const double q[4][2], double half, double H[9])
{
    const double x0 = q[0][0], y0 = q[0][1], x1 = q[1][0], y1 = q[1][1], x2 = q[2][0], y2 = q[2][1], x3 = q[3][0], y3 = q[3][1];
    const double dx1 = x1 - x2, dx2 = x3 - x2, sx = ((x0 - x1) + x2) - x3;
    const double dy1 = y1 - y2, dy2 = y3 - y2, sy = ((y0 - y1) + y2) - y3;
    const double den = dx1 * dy2 - dx2 * dy1;
    const double g = (sx * dy2 - dx2 * sy) / den;
    const double hh = (dx1 * sy - sx * dy1) / den;
    const double a = (x1 - x0) + g * x1, b = (x3 - x0) + hh * x3, c = x0;
    const double d = (y1 - y0) + g * y1, e = (y3 - y0) + hh * y3, f = y0;
    const double s = 1.0 / (2.0 * half);
    double M[9];
    M[0] = a * s; M[1] = -(b * s); M[2] = (a + b) * 0.5 + c;
    M[3] = d * s; M[4] = -(e * s); M[5] = (d + e) * 0.5 + f;
    M[6] = g * s; M[7] = -(hh * s); M[8] = (g + hh) * 0.5 + 1.0;
    const double inv = 1.0 / M[8];
#pragma unroll
    for (int i = 0; i < 8; i++) H[i] = M[i] * inv;
    H[8] = 1.0;
}

__device__ inline void rotate_vec_to_z(const double a[3], double R[9])
{
    double ax = a[0], ay = a[1], az = a[2];
    const double nrm = dsqrt(ax * ax + ay * ay + az * az);
    ax = ax / nrm;
    ay = ay / nrm;
    az = az / nrm;
    const double c = az;
    if (dabs(1.0 + c) < DBL_EPSILON) {
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = 0;
        R[0] = 1.0; R[4] = 1.0; R[8] = -1.0;
        return;
    }
    const double d = 1.0 / (1.0 + c);
    const double ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
    R[0] = -ax2 * d + 1.0; R[1] = -axay * d;       R[2] = -ax;
    R[3] = -axay * d;      R[4] = -ay2 * d + 1.0;  R[5] = -ay;
    R[6] = ax;             R[7] = ay;              R[8] = 1.0 - (ax2 + ay2) * d;
}

__device__ inline int compute_rotations(double j00, double j01, double j10, double j11, double p, double q, double R1[9], double R2[9])
{
    double Rv[9], v[3] = {p, q, 1.0};
    rotate_vec_to_z(v, Rv);
    const double rv00 = Rv[0], rv01 = Rv[3], rv02 = Rv[6], rv10 = Rv[1], rv11 = Rv[4], rv12 = Rv[7], rv20 = Rv[2], rv21 = Rv[5],
                 rv22 = Rv[8];
    const double b00 = rv00 - p * rv20, b01 = rv01 - p * rv21, b10 = rv10 - q * rv20, b11 = rv11 - q * rv21;
    const double dtinv = 1.0 / (b00 * b11 - b01 * b10);
    const double binv00 = dtinv * b11, binv01 = -dtinv * b01, binv10 = -dtinv * b10, binv11 = dtinv * b00;
    const double a00 = binv00 * j00 + binv01 * j10, a01 = binv00 * j01 + binv01 * j11;
    const double a10 = binv10 * j00 + binv11 * j10, a11 = binv10 * j01 + binv11 * j11;
    const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    const double gamma2 = 0.5 * (ata00 + ata11 + dsqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    if (!(gamma2 >= 0)) return 1;
    const double gamma = dsqrt(gamma2);
    if (dabs(gamma) < DBL_EPSILON) return 1;
    const double rt00 = a00 / gamma, rt01 = a01 / gamma, rt10 = a10 / gamma, rt11 = a11 / gamma;
    const double b0sq = -rt00 * rt00 - rt10 * rt10 + 1.0, b1sq = -rt01 * rt01 - rt11 * rt11 + 1.0;
    const double b0 = dsqrt(b0sq > 0 ? b0sq : 0.0);
    double b1 = dsqrt(b1sq > 0 ? b1sq : 0.0);
    const double sp = -rt00 * rt01 - rt10 * rt11;
    if (sp < 0) b1 = -b1;
    const double c0 = b1 * rt10 - b0 * rt11, c1 = b0 * rt01 - b1 * rt00, c2 = rt00 * rt11 - rt01 * rt10;
    const double rv[3][3] = {{rv00, rv01, rv02}, {rv10, rv11, rv12}, {rv20, rv21, rv22}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        R1[3 * i + 0] = rt00 * rv[i][0] + rt10 * rv[i][1] + b0 * rv[i][2];
        R1[3 * i + 1] = rt01 * rv[i][0] + rt11 * rv[i][1] + b1 * rv[i][2];
        R1[3 * i + 2] = c0 * rv[i][0] + c1 * rv[i][1] + c2 * rv[i][2];
        R2[3 * i + 0] = rt00 * rv[i][0] + rt10 * rv[i][1] + (-b0) * rv[i][2];
        R2[3 * i + 1] = rt01 * rv[i][0] + rt11 * rv[i][1] + (-b1) * rv[i][2];
        R2[3 * i + 2] = (-c0) * rv[i][0] + (-c1) * rv[i][1] + c2 * rv[i][2];
    }
    return 0;
}

__device__ inline void compute_translation(const double obj[4][2], const double img[4][2], const double R[9], double t[3])
{
    const double n = 4.0;
    double ATA00 = n, ATA02 = 0, ATA11 = n, ATA12 = 0, ATA20 = 0, ATA21 = 0, ATA22 = 0;
    double ATb0 = 0, ATb1 = 0, ATb2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double rx = R[0] * obj[i][0] + R[1] * obj[i][1];
        const double ry = R[3] * obj[i][0] + R[4] * obj[i][1];
        const double rz = R[6] * obj[i][0] + R[7] * obj[i][1];
        const double a2 = -img[i][0], b2 = -img[i][1];
        ATA02 = ATA02 + a2;
        ATA12 = ATA12 + b2;
        ATA20 = ATA20 + a2;
        ATA21 = ATA21 + b2;
        ATA22 = ATA22 + a2 * a2 + b2 * b2;
        const double bx = -a2 * rz - rx, by = -b2 * rz - ry;
        ATb0 = ATb0 + bx;
        ATb1 = ATb1 + by;
        ATb2 = ATb2 + a2 * bx + b2 * by;
    }
    const double detAInv = 1.0 / (ATA00 * ATA11 * ATA22 - ATA00 * ATA12 * ATA21 - ATA02 * ATA11 * ATA20);
    const double S00 = ATA11 * ATA22 - ATA12 * ATA21, S01 = ATA02 * ATA21, S02 = -ATA02 * ATA11;
    const double S10 = ATA12 * ATA20, S11 = ATA00 * ATA22 - ATA02 * ATA20, S12 = -ATA00 * ATA12;
    const double S20 = -ATA11 * ATA20, S21 = -ATA00 * ATA21, S22 = ATA00 * ATA11;
    t[0] = detAInv * (S00 * ATb0 + S01 * ATb1 + S02 * ATb2);
    t[1] = detAInv * (S10 * ATb0 + S11 * ATb1 + S12 * ATb2);
    t[2] = detAInv * (S20 * ATb0 + S21 * ATb1 + S22 * ATb2);
}

__device__ inline void rot2vec(const double R[9], double r[3])
{
    const double trace = R[0] + R[4] + R[8];
    const double w_norm = pm_acos((trace - 1.0) / 2.0);
    const double eps = (double)FLT_EPSILON;
    if (w_norm < eps) {
        r[0] = r[1] = r[2] = 0;
        return;
    }
    const double d = 1 / (2 * pm_sin(w_norm)) * w_norm;
    r[0] = d * (R[7] - R[5]);
    r[1] = d * (R[2] - R[6]);
    r[2] = d * (R[3] - R[1]);
}

__device__ inline float reproj_error(const double obj[4][2], const double img[4][2], const double R[9], const double t[3])
{
    float err = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double X = R[0] * obj[i][0] + R[1] * obj[i][1] + t[0];
        const double Y = R[3] * obj[i][0] + R[4] * obj[i][1] + t[1];
        const double Z = R[6] * obj[i][0] + R[7] * obj[i][1] + t[2];
        const double z = Z != 0 ? 1. / Z : 1.;
        const float dx = (float)(X * z) - (float)img[i][0], dy = (float)(Y * z) - (float)img[i][1];
        err += dx * dx + dy * dy;
    }
    return __builtin_sqrtf(err / (2.0f * 4));
}

// mobility.cpp:166-190 with the default ROI; returns 1 for degenerate points (rvec = tvec = 0)
__device__ inline int solve_pnp(const float vertices[4][2], const rmcv_pnp_config& cfg, double rvec[3], double tvec[3])
{
    const float hw = cfg.square_w / 2.0f, hhgt = cfg.square_h / 2.0f;
    const double obj[4][2] = {{-hw, hhgt}, {hw, hhgt}, {hw, -hhgt}, {-hw, -hhgt}}; // :175-180
    double img[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++) { // :182-185: image points 1, 2, 3, 0 (+ the zero ROI offset)
        const int o = (i + 1) & 3;
        float nx, ny;
        undistort_point(vertices[o][0] + 0.0f, vertices[o][1] + 0.0f, cfg.camera_matrix, cfg.dist, &nx, &ny);
        img[i][0] = nx;
        img[i][1] = ny;
    }
    rvec[0] = rvec[1] = rvec[2] = tvec[0] = tvec[1] = tvec[2] = 0;
    const float ddx = (float)obj[1][0] - (float)obj[0][0], ddy = (float)obj[1][1] - (float)obj[0][1];
    const double square_length = __builtin_sqrtf(ddx * ddx + ddy * ddy);
    {
        const double den = (img[1][0] - img[2][0]) * (img[3][1] - img[2][1]) - (img[3][0] - img[2][0]) * (img[1][1] - img[2][1]);
        if (!(dabs(den) > 0)) return 1;
    }
    double H[9];
    homography_from_square(img, square_length / 2.0, H);
#ifdef RMCV_PNP_DEBUG
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        printf("img %a %a %a %a %a %a %a %a\n", img[0][0], img[0][1], img[1][0], img[1][1], img[2][0], img[2][1], img[3][0], img[3][1]);
        printf("sq %a H %a %a %a %a %a %a %a %a\n", square_length, H[0], H[1], H[2], H[3], H[4], H[5], H[6], H[7]);
    }
#endif
    const double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double Ra[9], Rb[9], ta[3], tb[3];
    if (compute_rotations(j00, j01, j10, j11, H[2], H[5], Ra, Rb)) return 1;
    compute_translation(obj, img, Ra, ta);
    compute_translation(obj, img, Rb, tb);
    const float ea = reproj_error(obj, img, Ra, ta), eb = reproj_error(obj, img, Rb, tb);
    const bool first_a = !(ea > eb);
    double Rsel[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Rsel[i] = first_a ? Ra[i] : Rb[i];
    rot2vec(Rsel, rvec);
#pragma unroll
    for (int i = 0; i < 3; i++) tvec[i] = first_a ? ta[i] : tb[i];
    return 0;
}

// one wavefront per frame, one lane per armour; poses[frame][armour] = rvec[3] | tvec[3] | position[3]
__global__ __launch_bounds__(64) void k_pnp(const rmcv_armour* __restrict__ armours, const int32_t* __restrict__ n_armours,
                                           int max_armours, const rmcv_pnp_config* __restrict__ cfg_p,
                                           const double* __restrict__ base2gripper, double* __restrict__ poses)
{
    const int f = blockIdx.x;
    const rmcv_pnp_config cfg = *cfg_p;
    int n = n_armours[f];
    n = n > max_armours ? max_armours : n;
    const double* B = base2gripper + (int64_t)f * 16;
    for (int i = threadIdx.x; i < n; i += 64) {
        const rmcv_armour* a = armours + (int64_t)f * max_armours + i;
        float v[4][2];
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k][0] = a->vertices[k][0]; v[k][1] = a->vertices[k][1]; }
        double r[3], t[3];
        solve_pnp(v, cfg, r, t);
        // main.cpp:186-192: world = h_base2gripper * (h_gripper2camera * [tvec; 1])
        const double cam[4] = {t[0], t[1], t[2], 1.0};
        double mid[4], out[3];
#pragma unroll
        for (int k = 0; k < 4; k++)
            mid[k] = cfg.gripper2camera[4 * k] * cam[0] + cfg.gripper2camera[4 * k + 1] * cam[1] + cfg.gripper2camera[4 * k + 2] * cam[2] +
                     cfg.gripper2camera[4 * k + 3] * cam[3];
#pragma unroll
        for (int k = 0; k < 3; k++) out[k] = B[4 * k] * mid[0] + B[4 * k + 1] * mid[1] + B[4 * k + 2] * mid[2] + B[4 * k + 3] * mid[3];
        double* o = poses + ((int64_t)f * max_armours + i) * 9;
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = r[k]; o[3 + k] = t[k]; o[6 + k] = out[k]; }
    }
}

hipError_t launch_pnp(const Geom& g, const Bufs& b, const Limits& lim, hipStream_t s)
{
    return launch(k_pnp, dim3(g.n_frames), dim3(64), 0, s, b.armours, b.n_armours, lim.max_armours, b.pnp_cfg, b.base2gripper,
                       b.poses);
}

} // namespace rmcv
