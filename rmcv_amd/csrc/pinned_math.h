/*
 * pinned_math.h -- the handful of transcendental functions the detection path
 * needs, written once in plain IEEE-754 arithmetic (+ - * / and bit moves only)
 * so that the SAME source gives the SAME bits on the host (gcc, used by the
 * CPU oracle) and on the device (hipcc, gfx950).  Both sides must be compiled
 * with -ffp-contract=off and without fast-math; nothing here may be replaced
 * by a library call.
 *
 * Why it exists: the reference leaves these to the platform libm
 *   - cv::fitEllipseDirect        -> atan2 (double)        (src/objdetect.cpp:68)
 *   - cv::RotatedRect::points     -> sin, cos (double)     (src/core.cpp:268)
 *   - rm::filter_armours          -> atan2 on floats       (src/objdetect.cpp:137)
 *   - rm::utils::ExtendCord       -> atan2/sin/cos floats  (src/core.cpp:335-337)
 * and a GPU has no glibc.  Every double result on this path is narrowed to
 * float before it is used, so a <=1 ulp (double) difference from glibc changes
 * an output bit with probability ~2^-29 per call; tests/test_pinned_math.py
 * measures the agreement with the host libm.
 *
 * Method: classic Cody-Waite reduction by pi/2 (three 33/33/53-bit pieces) and
 * the well-known degree-13/14 minimax kernels on [-pi/4, pi/4]; arctangent by
 * the usual 4-breakpoint reduction + odd/even split degree-11 polynomial in
 * x^2.  Coefficients were re-verified against mpmath (rel. error < 5e-18).
 * Float variants evaluate in double and round once (correctly rounded float
 * results except with probability ~2^-29).
 */
#ifndef RMCV_PINNED_MATH_H
#define RMCV_PINNED_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PM_FN __host__ __device__ static inline
#else
#define PM_FN static inline
#endif

PM_FN double pm_fabs(double x) { return x < 0 ? -x : (x == 0 ? 0.0 : x); }

PM_FN double pm_hi_word_only(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    u &= 0xFFFFFFFF00000000ull;
    __builtin_memcpy(&x, &u, 8);
    return x;
}

/* sin on [-pi/4, pi/4]; y is the tail of x */
PM_FN double pm_ksin(double x, double y, int have_tail)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (!have_tail) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* cos on [-pi/4, pi/4]; y is the tail of x */
PM_FN double pm_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ax = pm_fabs(x);
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
    {
        double qx = (ax > 0.78125) ? 0.28125 : pm_hi_word_only(ax * 0.25);
        double hz = 0.5 * z - qx;
        double a = 1.0 - qx;
        return a - (hz - (z * r - x * y));
    }
}

/* reduce x to r+t with |r| <= pi/4 (+eps); returns quadrant n mod 4.
 * Accurate for |x| < ~1e5 (n*p1 and n*p2 exact while |n| < 2^20). */
PM_FN int pm_rem_pio2(double x, double* r_hi, double* r_lo)
{
    const double invpio2 = 0.6366197723675814;
    const double p1 = 1.5707963267341256;       /* first 33 bits of pi/2 */
    const double p2 = 6.077100506303966e-11;    /* next 33 bits          */
    const double p3 = 2.0222662487959506e-21;   /* the rest, as a double */
    double fn = x * invpio2;
    int n = (int)(fn < 0 ? fn - 0.5 : fn + 0.5);
    double dn = (double)n;
    double r = x - dn * p1;
    double w = dn * p2;
    double hi = r - w;
    double lo = (r - hi) - w;
    double t = dn * p3;
    double hi2 = hi - t;
    double lo2 = ((hi - hi2) - t) + lo;
    *r_hi = hi2;
    *r_lo = lo2;
    return n & 3;
}

PM_FN double pm_sin(double x)
{
    double r, t;
    if (pm_fabs(x) <= 0.7853981633974483) return pm_ksin(x, 0.0, 0);
    switch (pm_rem_pio2(x, &r, &t)) {
        case 0: return pm_ksin(r, t, 1);
        case 1: return pm_kcos(r, t);
        case 2: return -pm_ksin(r, t, 1);
        default: return -pm_kcos(r, t);
    }
}

PM_FN double pm_cos(double x)
{
    double r, t;
    if (pm_fabs(x) <= 0.7853981633974483) return pm_kcos(x, 0.0);
    switch (pm_rem_pio2(x, &r, &t)) {
        case 0: return pm_kcos(r, t);
        case 1: return -pm_ksin(r, t, 1);
        case 2: return -pm_kcos(r, t);
        default: return pm_ksin(r, t, 1);
    }
}

PM_FN double pm_atan(double x)
{
    const double hi0 = 4.63647609000806093515e-01, lo0 = 2.26987774529616870924e-17; /* atan(.5) */
    const double hi1 = 7.85398163397448278999e-01, lo1 = 3.06161699786838301793e-17; /* atan(1)  */
    const double hi2 = 9.82793723247329054082e-01, lo2 = 1.39033110312309984516e-17; /* atan(1.5)*/
    const double hi3 = 1.57079632679489655800e+00, lo3 = 6.12323399573676603587e-17; /* atan(inf)*/
    const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
                 a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
                 a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
                 a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
                 a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
                 a10 = 1.62858201153657823623e-02;
    int neg = x < 0;
    double ax = pm_fabs(x), hi = 0, lo = 0, z, w, s1, s2, res;
    int id;
    if (ax != ax) return x; /* NaN */
    if (ax >= 7.37869762948382064640e+19) { /* 2^66 */
        res = hi3 + lo3;
        return neg ? -res : res;
    }
    if (ax < 0.4375) {
        if (ax < 3.725290298461914e-09) return x; /* 2^-28: atan(x) = x */
        id = -1;
    } else if (ax < 1.1875) {
        if (ax < 0.6875) { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); hi = hi0; lo = lo0; }
        else             { id = 1; ax = (ax - 1.0) / (ax + 1.0);       hi = hi1; lo = lo1; }
    } else {
        if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); hi = hi2; lo = lo2; }
        else             { id = 3; ax = -1.0 / ax;                     hi = hi3; lo = lo3; }
    }
    z = ax * ax;
    w = z * z;
    s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) res = ax - ax * (s1 + s2);
    else        res = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -res : res;
}

/* atan2 for finite arguments (infinities/NaN are not produced on this path;
 * they are still mapped to something sensible). */
PM_FN double pm_atan2(double y, double x)
{
    const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16;
    const double pio2 = 1.5707963267948965580e+00;
    double z;
    if (x != x || y != y) return x + y;
    if (y == 0.0) {
        /* sign of zero is dropped: +-0/x -> 0 for x>=0, pi for x<0 */
        return (x < 0) ? pi : 0.0;
    }
    if (x == 0.0) return (y < 0) ? -pio2 : pio2;
    {
        double ay = pm_fabs(y), ax = pm_fabs(x);
        double q = ay / ax;
        z = pm_atan(q);
        if (x > 0) return (y < 0) ? -z : z;
        z = pi - (z - pi_lo);
        return (y < 0) ? -z : z;
    }
}

/* arccosine (IPPE's rotation-vector conversion, the solve_PnP row): the classic rational kernel R(z) ~ (asin(x) - x) / x^3
 * on |x| <= 0.5 and the sqrt reductions outside, |error| < 1 ulp.  sqrt is IEEE (correctly rounded) on host and device. */
PM_FN double pm_acos(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    double ax = pm_fabs(x), z, p, q, r, s, w;
    if (x != x) return x;
    if (ax >= 1.0) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x); /* NaN */
    }
    if (ax < 0.5) {
        if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo; /* 2^-57 */
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0) {
        z = (1.0 + x) * 0.5;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = __builtin_sqrt(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    {
        double df, c;
        z = (1.0 - x) * 0.5;
        s = __builtin_sqrt(z);
        df = pm_hi_word_only(s);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0 * (df + w);
    }
}

/* float variants: one evaluation in double, one rounding */
PM_FN float pm_atan2f(float y, float x) { return (float)pm_atan2((double)y, (double)x); }
PM_FN float pm_sinf(float x) { return (float)pm_sin((double)x); }
PM_FN float pm_cosf(float x) { return (float)pm_cos((double)x); }

/* fmod(x, 180) for 0 <= x < 720 -- each subtraction is exact (Sterbenz) */
PM_FN double pm_fmod180(double x)
{
    while (x >= 180.0) x -= 180.0;
    return x;
}

#endif /* RMCV_PINNED_MATH_H */
