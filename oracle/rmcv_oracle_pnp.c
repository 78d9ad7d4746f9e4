/*
 * rmcv_oracle_pnp.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), SURVEY 8f-3: per-armour pose
 *   rm::solve_PnP                /root/reference/src/mobility.cpp:166-190   (cv::solvePnP, SOLVEPNP_IPPE_SQUARE)
 *   camera -> world position     /root/reference/executable/main.cpp:183-192
 *   camera constants             /root/reference/executable/main.cpp:7-19
 *
 * [OCV] cv::solvePnP(IPPE_SQUARE) = cv::undistortPoints (5 fixed-point iterations, result stored as float because the
 * image points are Point2f) followed by IPPE::PoseSolver::solveSquare (Collins & Bartoli, "Infinitesimal Plane-based Pose
 * Estimation", IJCV 2014): homography of the square, its Jacobian at the origin, the two rotations, each one's
 * least-squares translation, the pose with the smaller reprojection error first.  OpenCV is an un-vendored dependency:
 * the structure follows its implementation as recalled, the homography's closed form and the order of the floating-point
 * operations are this file's own.  PARITY UNPINNED against real OpenCV (see rmcv_oracle.h); tests/test_oracle_pnp.py
 * checks the solver against poses it was not told (forward projection with numpy).
 */
#include <float.h>
#include <math.h>
#include <string.h>
#ifdef ORC_PNP_DEBUG
#include <stdio.h>
#endif

#include "../rmcv_amd/csrc/pinned_math.h"
#include "rmcv_oracle.h"

static double p_acos(double x) { return orc_get_math_mode() ? acos(x) : pm_acos(x); }
static double p_sin(double x) { return orc_get_math_mode() ? sin(x) : pm_sin(x); }

/* executable/main.cpp:7-19: every literal carries an `f` suffix, i.e. it is a float widened to double */
void orc_default_pnp_config(orc_pnp_config* c)
{
    const float K[9] = {1782.672144409928f, 0.0f, 598.8983414505224f, 0.0f, 1783.860175007369f, 523.4209809658056f, 0.0f, 0.0f, 1.0f};
    const float D[5] = {-0.03436366268485048f, 0.1953669264956857f, 0.0001485060439399386f, -0.003814875777013483f,
                        -0.3181808766352414f};
    const float G[16] = {0.0007941130268316332f, 0.009683274185178004f, -0.9999528006788897f, -27.25811584661768f,
                         0.9989588796104363f, 0.04560298009571095f, 0.001234930707386894f, -51.46996511920027f,
                         0.04561278583864914f, -0.9989127101040636f, -0.009636978810429797f, 77.11760876626687f,
                         0.0f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < 9; i++) c->camera_matrix[i] = K[i];
    for (int i = 0; i < 5; i++) c->dist[i] = D[i];
    for (int i = 0; i < 16; i++) c->gripper2camera[i] = G[i];
    c->square_w = 27.0f; /* main.cpp:184 {27, 27} */
    c->square_h = 27.0f;
}

/* [OCV] undistortPoints(src, dst, K, dist) without R/P: TermCriteria(MAX_ITER, 5, 0.01) -> exactly 5 iterations of
 * x <- (x0 - tangential(x)) / radial(x); k = (k1, k2, p1, p2, k3), the higher coefficients are zero.  dst has the type
 * of src (CV_32FC2). */
static void undistort_point(float u, float v, const double* K, const double* k, float* ox, float* oy)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { /* OpenCV gives up on the point and returns the undistorted guess */
            x = ((double)u - cx) * ifx;
            y = ((double)v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    { /* the narrowing IS the semantics (dst is CV_32FC2): keep it out of the optimiser's reach */
        volatile float fx32 = (float)x, fy32 = (float)y;
        *ox = fx32;
        *oy = fy32;
    }
}

/* homography from the canonical square (-h, h), (h, h), (h, -h), (-h, -h) to the four (normalised) image points q0..q3,
 * scaled so that H[8] = 1.  Closed form: unit square -> quadrilateral (projective mapping through the two vanishing
 * coefficients g, h), composed with the affine map square -> unit square. */
static void homography_from_square(const double q[4][2], double half, double H[9])
{
    const double x0 = q[0][0], y0 = q[0][1], x1 = q[1][0], y1 = q[1][1], x2 = q[2][0], y2 = q[2][1], x3 = q[3][0], y3 = q[3][1];
    const double dx1 = x1 - x2, dx2 = x3 - x2, sx = ((x0 - x1) + x2) - x3;
    const double dy1 = y1 - y2, dy2 = y3 - y2, sy = ((y0 - y1) + y2) - y3;
    const double den = dx1 * dy2 - dx2 * dy1;
    const double g = (sx * dy2 - dx2 * sy) / den;
    const double hh = (dx1 * sy - sx * dy1) / den;
    /* unit square (u, v) -> image: [a b c; d e f; g hh 1] */
    const double a = (x1 - x0) + g * x1, b = (x3 - x0) + hh * x3, c = x0;
    const double d = (y1 - y0) + g * y1, e = (y3 - y0) + hh * y3, f = y0;
    /* (X, Y) -> (u, v) = ((X + half) / (2 half), (half - Y) / (2 half)) */
    const double s = 1.0 / (2.0 * half);
    double M[9];
    M[0] = a * s; M[1] = -(b * s); M[2] = (a + b) * 0.5 + c;
    M[3] = d * s; M[4] = -(e * s); M[5] = (d + e) * 0.5 + f;
    M[6] = g * s; M[7] = -(hh * s); M[8] = (g + hh) * 0.5 + 1.0;
    const double inv = 1.0 / M[8];
    for (int i = 0; i < 8; i++) H[i] = M[i] * inv;
    H[8] = 1.0;
}

/* IPPE: rotation that takes a to the z axis */
static void rotate_vec_to_z(const double a[3], double R[9])
{
    double ax = a[0], ay = a[1], az = a[2];
    const double nrm = sqrt(ax * ax + ay * ay + az * az);
    ax = ax / nrm;
    ay = ay / nrm;
    az = az / nrm;
    const double c = az;
    if (fabs(1.0 + c) < DBL_EPSILON) {
        memset(R, 0, 9 * sizeof(double));
        R[0] = 1.0; R[4] = 1.0; R[8] = -1.0;
        return;
    }
    const double d = 1.0 / (1.0 + c);
    const double ax2 = ax * ax, ay2 = ay * ay, axay = ax * ay;
    R[0] = -ax2 * d + 1.0; R[1] = -axay * d;       R[2] = -ax;
    R[3] = -axay * d;      R[4] = -ay2 * d + 1.0;  R[5] = -ay;
    R[6] = ax;             R[7] = ay;              R[8] = 1.0 - (ax2 + ay2) * d;
}

/* IPPE: the two rotations from the homography's Jacobian J at the origin and the image of the origin (p, q) */
static int compute_rotations(double j00, double j01, double j10, double j11, double p, double q, double R1[9], double R2[9])
{
    double Rv[9], v[3] = {p, q, 1.0};
    rotate_vec_to_z(v, Rv);
    /* Rv = Rv.t() */
    const double rv00 = Rv[0], rv01 = Rv[3], rv02 = Rv[6], rv10 = Rv[1], rv11 = Rv[4], rv12 = Rv[7], rv20 = Rv[2], rv21 = Rv[5],
                 rv22 = Rv[8];
    const double b00 = rv00 - p * rv20, b01 = rv01 - p * rv21, b10 = rv10 - q * rv20, b11 = rv11 - q * rv21;
    const double dtinv = 1.0 / (b00 * b11 - b01 * b10);
    const double binv00 = dtinv * b11, binv01 = -dtinv * b01, binv10 = -dtinv * b10, binv11 = dtinv * b00;
    const double a00 = binv00 * j00 + binv01 * j10, a01 = binv00 * j01 + binv01 * j11;
    const double a10 = binv10 * j00 + binv11 * j10, a11 = binv10 * j01 + binv11 * j11;
    /* largest singular value of A */
    const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    const double gamma2 = 0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    if (!(gamma2 >= 0)) return 1;
    const double gamma = sqrt(gamma2);
    if (fabs(gamma) < DBL_EPSILON) return 1;
    const double rt00 = a00 / gamma, rt01 = a01 / gamma, rt10 = a10 / gamma, rt11 = a11 / gamma;
    double b0sq = -rt00 * rt00 - rt10 * rt10 + 1.0, b1sq = -rt01 * rt01 - rt11 * rt11 + 1.0;
    double b0 = sqrt(b0sq > 0 ? b0sq : 0.0), b1 = sqrt(b1sq > 0 ? b1sq : 0.0);
    const double sp = -rt00 * rt01 - rt10 * rt11;
    if (sp < 0) b1 = -b1;
    const double c0 = b1 * rt10 - b0 * rt11, c1 = b0 * rt01 - b1 * rt00, c2 = rt00 * rt11 - rt01 * rt10;
    const double rv[3][3] = {{rv00, rv01, rv02}, {rv10, rv11, rv12}, {rv20, rv21, rv22}};
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

/* IPPE: least-squares translation for a given rotation (normal equations of A t = b, closed-form 3x3 inverse) */
static void compute_translation(const double obj[4][2], const double img[4][2], const double R[9], double t[3])
{
    const double n = 4.0;
    double ATA00 = n, ATA02 = 0, ATA11 = n, ATA12 = 0, ATA20 = 0, ATA21 = 0, ATA22 = 0;
    double ATb0 = 0, ATb1 = 0, ATb2 = 0;
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

/* IPPE: rotation matrix -> rotation vector */
static void rot2vec(const double R[9], double r[3])
{
    const double trace = R[0] + R[4] + R[8];
    const double w_norm = p_acos((trace - 1.0) / 2.0);
    const double eps = (double)FLT_EPSILON;
    if (w_norm < eps) {
        r[0] = r[1] = r[2] = 0;
        return;
    }
    const double d = 1 / (2 * p_sin(w_norm)) * w_norm;
    r[0] = d * (R[7] - R[5]);
    r[1] = d * (R[2] - R[6]);
    r[2] = d * (R[3] - R[1]);
}

/* IPPE: RMS reprojection error in normalised coordinates, accumulated in float like OpenCV's evalReprojError */
static float reproj_error(const double obj[4][2], const double img[4][2], const double R[9], const double t[3])
{
    float err = 0;
    for (int i = 0; i < 4; i++) {
        const double X = R[0] * obj[i][0] + R[1] * obj[i][1] + t[0];
        const double Y = R[3] * obj[i][0] + R[4] * obj[i][1] + t[1];
        const double Z = R[6] * obj[i][0] + R[7] * obj[i][1] + t[2];
        const double z = Z != 0 ? 1. / Z : 1.;
        const float dx = (float)(X * z) - (float)img[i][0], dy = (float)(Y * z) - (float)img[i][1];
        err += dx * dx + dy * dy;
    }
    return sqrtf(err / (2.0f * 4));
}

/* rm::solve_PnP(armour.vertices, cammat, discof, {27, 27}) -- mobility.cpp:166-190 with the default ROI (0,0,0,0).
 * Returns 0, or 1 when the four points are degenerate (then rvec/tvec are zero). */
int orc_solve_pnp(const float vertices[4][2], const orc_pnp_config* cfg, double rvec[3], double tvec[3])
{
    /* mobility.cpp:175-185: object corners (-w/2, h/2), (w/2, h/2), (w/2, -h/2), (-w/2, -h/2); image points 1, 2, 3, 0 */
    const float hw = cfg->square_w / 2.0f, hhgt = cfg->square_h / 2.0f;
    const double obj[4][2] = {{-hw, hhgt}, {hw, hhgt}, {hw, -hhgt}, {-hw, -hhgt}};
    static const int order[4] = {1, 2, 3, 0};
    double img[4][2];
    for (int i = 0; i < 4; i++) {
        float nx, ny;
        undistort_point(vertices[order[i]][0] + 0.0f, vertices[order[i]][1] + 0.0f, cfg->camera_matrix, cfg->dist, &nx, &ny);
        img[i][0] = nx;
        img[i][1] = ny;
    }
    rvec[0] = rvec[1] = rvec[2] = tvec[0] = tvec[1] = tvec[2] = 0;
    /* solveSquare: side length from the first two object points (float arithmetic) */
    const float ddx = (float)obj[1][0] - (float)obj[0][0], ddy = (float)obj[1][1] - (float)obj[0][1];
    const double square_length = sqrtf(ddx * ddx + ddy * ddy);
    double H[9];
    {
        const double den = (img[1][0] - img[2][0]) * (img[3][1] - img[2][1]) - (img[3][0] - img[2][0]) * (img[1][1] - img[2][1]);
        if (!(fabs(den) > 0)) return 1;
    }
    homography_from_square(img, square_length / 2.0, H);
#ifdef ORC_PNP_DEBUG
    printf("img %a %a %a %a %a %a %a %a\n", img[0][0], img[0][1], img[1][0], img[1][1], img[2][0], img[2][1], img[3][0], img[3][1]);
    printf("sq %a H %a %a %a %a %a %a %a %a\n", square_length, H[0], H[1], H[2], H[3], H[4], H[5], H[6], H[7]);
#endif
    const double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double Ra[9], Rb[9], ta[3], tb[3];
    if (compute_rotations(j00, j01, j10, j11, H[2], H[5], Ra, Rb)) return 1;
    compute_translation(obj, img, Ra, ta);
    compute_translation(obj, img, Rb, tb);
    const float ea = reproj_error(obj, img, Ra, ta), eb = reproj_error(obj, img, Rb, tb);
    const int first_a = !(ea > eb); /* the poses are swapped only when the first one is strictly worse */
    rot2vec(first_a ? Ra : Rb, rvec);
    memcpy(tvec, first_a ? ta : tb, 3 * sizeof(double));
    return 0;
}

/* main.cpp:186-192: world = h_base2gripper * (h_gripper2camera * [tvec; 1]); row-major 4x4, dot products left to right */
void orc_armour_position(const double tvec[3], const double base2gripper[16], const double gripper2camera[16], double pos[3])
{
    const double cam[4] = {tvec[0], tvec[1], tvec[2], 1.0};
    double mid[4], out[4];
    for (int i = 0; i < 4; i++)
        mid[i] = gripper2camera[4 * i] * cam[0] + gripper2camera[4 * i + 1] * cam[1] + gripper2camera[4 * i + 2] * cam[2] +
                 gripper2camera[4 * i + 3] * cam[3];
    for (int i = 0; i < 4; i++)
        out[i] = base2gripper[4 * i] * mid[0] + base2gripper[4 * i + 1] * mid[1] + base2gripper[4 * i + 2] * mid[2] +
                 base2gripper[4 * i + 3] * mid[3];
    pos[0] = out[0];
    pos[1] = out[1];
    pos[2] = out[2];
}

/* the pose part of the loop body of main.cpp:178-196 for n armours of one frame */
void orc_locate_armours(const orc_armour* armours, int n, const orc_pnp_config* cfg, const double base2gripper[16],
                        double* rvecs /* n*3 */, double* tvecs /* n*3 */, double* positions /* n*3 */)
{
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < n; i++) {
        double r[3], t[3], p[3];
        orc_solve_pnp(armours[i].vertices, cfg, r, t);
        orc_armour_position(t, base2gripper ? base2gripper : eye, cfg->gripper2camera, p);
        for (int k = 0; k < 3; k++) {
            if (rvecs) rvecs[3 * i + k] = r[k];
            if (tvecs) tvecs[3 * i + k] = t[k];
            if (positions) positions[3 * i + k] = p[k];
        }
    }
}
