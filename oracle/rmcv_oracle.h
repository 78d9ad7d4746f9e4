/*
 * rmcv_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's per-frame armour-detection path
 *   rm::extract_color      /root/reference/src/imgproc.cpp:50-75
 *   rm::filter_lightblobs  /root/reference/src/objdetect.cpp:55-87
 *   rm::filter_armours     /root/reference/src/objdetect.cpp:114-166
 *   rm::lightblob ctor     /root/reference/src/core.cpp:9-19, 265-283
 *   rm::armour ctor        /root/reference/src/core.cpp:21-49, 285-404
 * including the OpenCV (>= 4.8.0, vcpkg.json:28-35) primitives those call.
 *
 * PARITY STATUS: **parity unpinned** against real OpenCV.  OpenCV is an
 * un-vendored third-party dependency (not under /root/reference, not installed
 * here) and the reference ships no tests, fixtures or golden vectors
 * (SURVEY.md section 4, 8c).  Integer stages (threshold, morphology, contours,
 * contourArea) restate fully specified published behaviour; fitEllipseDirect
 * restates the published algorithm with an operation order documented in
 * rmcv_oracle.c.  What pins this file is tests/golden/ (hand-derived KATs).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library -- never the product path (rmcv_amd/).
 */
#ifndef RMCV_ORACLE_H
#define RMCV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rm::camp, include/core.h:20-23 */
enum { ORC_CAMP_RED = 0, ORC_CAMP_BLUE = 1, ORC_CAMP_GUIDELIGHT = 2, ORC_CAMP_NEUTRAL = -1 };
/* morphology selector: the snapshot does CLOSE; the older API did a dilate only */
enum { ORC_MORPH_NONE = 0, ORC_MORPH_DILATE = 1, ORC_MORPH_CLOSE = 2 };

typedef struct { int32_t x, y; } orc_point;                      /* cv::Point        */
typedef struct { float cx, cy, w, h, angle; } orc_rrect;          /* cv::RotatedRect  */
typedef struct {                                                  /* rm::lightblob, include/core.h:89-99 */
    float   angle;
    int32_t target;
    float   center[2];
    float   vertices[4][2];
    float   size[2];            /* (width=min, height=max) */
} orc_lightblob;                                                  /* 56 bytes */
typedef struct {                                                  /* rm::armour PODs, include/core.h:110-112 */
    float   icon[4][2];
    float   vertices[4][2];
    float   bbox[4];            /* x, y, width, height */
    int32_t blob_i, blob_j;     /* indices into the positive light-blob list */
} orc_armour;                                                     /* 88 bytes */

typedef struct {                /* literals of executable/main.cpp:172-176 are the defaults */
    int32_t camp;               /* enemy colour                          (CAMP_BLUE) */
    int32_t lower_bound;        /* inRange lower bound                   (80)        */
    int32_t morph;              /* ORC_MORPH_*                           (CLOSE)     */
    float   tilt_max;           /*                                       (70)        */
    float   ratio_lo, ratio_hi; /* range<float>                          (1.5, 80)   */
    double  area_lo, area_hi;   /* range<double>                         (10, 99999) */
    float   angle_diff_max;     /*                                       (12)        */
    float   shear_max;          /*                                       (22)        */
    float   length_ratio_max;   /*                                       (0.4)       */
    int32_t _pad;
} orc_params;

/* 0 = pinned_math.h (the parity contract with the GPU), 1 = host libm (what the
 * reference itself would link).  tests compare the two. */
void orc_set_math_mode(int mode);
int  orc_get_math_mode(void);
/* which functions the reference's unqualified abs / atan2 / sin / cos on floats resolve to -- bit 0: abs -> int abs(int),
 * bit 1: atan2 / sin / cos -> the double functions; 0 (default): the float overloads */
void orc_set_overload_mode(int mode);
int  orc_get_overload_mode(void);
float orc_abs_ov(float x);

void orc_default_params(orc_params* p);

/* imgproc.cpp:52-69 : binary = close3x3(inRange(sat(chA - chB), lb, 255)) */
int orc_extract_binary(const uint8_t* bgr, int w, int h, int stride, int camp, int lower_bound, int morph,
                       uint8_t* binary /* h*w, 0/255 */);
void orc_dilate3x3(const uint8_t* in, uint8_t* out, int w, int h);
void orc_erode3x3(const uint8_t* in, uint8_t* out, int w, int h);

/* imgproc.cpp:71-72 : cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE).
 * offs has n+1 entries; returns 0, or -2 if a capacity was exceeded (counts
 * are still the required sizes). */
int orc_find_contours(const uint8_t* binary, int w, int h, orc_point* pts, int cap_pts, int32_t* offs,
                      int cap_contours, int32_t* n_contours, int32_t* n_points);

double orc_contour_area(const orc_point* pts, int n);                   /* cv::contourArea */
int    orc_fit_ellipse_direct(const orc_point* pts, int n, orc_rrect* out); /* cv::fitEllipseDirect; returns
                                 0 direct solution, 1 fell back to the general (LIN) fit */
void   orc_rrect_points(const orc_rrect* r, float pt[4][2]);            /* cv::RotatedRect::points */
void   orc_make_lightblob(const orc_rrect* box, int camp, orc_lightblob* out); /* core.cpp:9-19 */
void   orc_make_armour(const orc_lightblob* a, const orc_lightblob* b, orc_armour* out); /* core.cpp:21-49 */

/* objdetect.cpp:55-87.  neg_idx receives the contour indices of the "negative" list. */
int orc_filter_lightblobs(const orc_point* pts, const int32_t* offs, int n_contours, float tilt_max,
                          float ratio_lo, float ratio_hi, double area_lo, double area_hi, int enemy,
                          orc_lightblob* blobs, int cap_blobs, int32_t* n_blobs, int32_t* blob_src,
                          int32_t* neg_idx, int32_t* n_neg, orc_rrect* ellipses /* optional, per positive */);

/* objdetect.cpp:114-166 */
int orc_filter_armours(const orc_lightblob* blobs, int n, float angle_diff_max, float shear_max,
                       float length_ratio_max, int enemy, orc_armour* out, int cap, int32_t* n_out);

/* the whole per-frame path, main.cpp:172-176; any output pointer may be NULL */
int orc_detect_frame(const uint8_t* bgr, int w, int h, int stride, const orc_params* p, uint8_t* binary,
                     orc_point* pts, int cap_pts, int32_t* offs, int cap_contours, int32_t* n_contours,
                     orc_lightblob* blobs, int cap_blobs, int32_t* n_blobs, orc_armour* armours, int cap_armours,
                     int32_t* n_armours);


/* ------------------------------------------------------------------------------------------------
 * "Next" row SURVEY 8f-1 / BASELINE config 5: icon ROI rectification + linear SVM digit classifier,
 *   rm::affine_correction            /root/reference/src/imgproc.cpp:9-35
 *   rm::utils::flatten_image         /root/reference/src/core.cpp:202-216
 *   svm->predict                     /root/reference/executable/main.cpp:180-181
 *   model shape                      /root/reference/executable/svm/optimizer.cpp:9,16-19  (7-class linear C_SVC)
 * [OCV] pieces (cvRound, boundingRect on int points, getAffineTransform's LU solve, warpAffine's and resize's 8-bit
 * fixed-point bilinear, the one-vs-one vote) restate OpenCV 4.8 as recalled -- parity unpinned like the rest. */
#define ORC_ICON_SIDE 20
#define ORC_SVM_FEATURES (ORC_ICON_SIDE * ORC_ICON_SIDE * 3)

/* icon: the armour's 4 icon vertices, clamped IN PLACE to the frame like the reference does (imgproc.cpp:11-15).
 * out: 20x20 BGR u8.  Returns 0, or 1 when the ROI is degenerate (then out is zero-filled). */
int orc_affine_correction(const uint8_t* bgr, int w, int h, int stride, float icon[4][2], uint8_t* out);
/* 7-class one-vs-one linear C_SVC: weights[n_df][n_feat] (n_df = n_class*(n_class-1)/2), rho[n_df], labels[n_class] */
int orc_svm_predict(const float* features, int n_feat, const float* weights, const double* rho, const int32_t* labels, int n_class);
/* main.cpp:178-181 for every armour of a frame: identity[i] = predict(flatten(affine_correction(icon))) */
void orc_classify_armours(const uint8_t* bgr, int w, int h, int stride, orc_armour* armours, int n, const float* weights,
                          const double* rho, const int32_t* labels, int n_class, int32_t* identity, uint8_t* icons /* n*1200 or NULL */);

/* ------------------------------------------------------------------------------------------------
 * "Next" row SURVEY 8f-2 (rmcv_oracle_legacy.c): the legacy per-contour matcher
 *   rm::MatchLightBlob /root/reference/src/objdetect.cpp:9-28, rm::FindLightBlobs :30-53, rm::LightBlobOverlap :89-112
 * with [OCV] cv::minAreaRect (convexHull + rotating calipers), boundingRect(int points), mean(ROI). */
int  orc_convex_hull(const orc_point* pts, int n, int32_t* hull_idx);        /* clockwise=false; indices into pts */
int  orc_convex_hull_pruned(const orc_point* pts, int n, int32_t* hull_idx); /* test cross-check of the kernel's column pruning */
void orc_min_area_rect(const orc_point* pts, int n, orc_rrect* box);
int  orc_match_lightblob(const orc_point* pts, int n, float min_ratio, float max_ratio, float tilt_angle, float min_area,
                         float max_area, int fit_ellipse, orc_rrect* box_out); /* 1 = matched */
void orc_bounding_rect(const orc_point* pts, int n, int32_t rect[4]);
int  orc_camp_from_mean(const uint8_t* bgr, int stride, const int32_t rect[4]);
int  orc_find_lightblobs(const uint8_t* bgr, int w, int h, int stride, const orc_point* pts, const int32_t* offs,
                         int n_contours, float min_ratio, float max_ratio, float tilt_angle, float min_area, float max_area,
                         int fit_ellipse, orc_lightblob* blobs, int cap_blobs, int32_t* n_blobs, int32_t* blob_src,
                         orc_rrect* boxes /* optional, per blob */);
int  orc_lightblob_overlap(const orc_lightblob* blobs, int n, int left, int right); /* 1/0; -1: right == n (UB in the reference) */

/* ------------------------------------------------------------------------------------------------
 * "Next" row SURVEY 8f-3 (rmcv_oracle_pnp.c): per-armour pose
 *   rm::solve_PnP /root/reference/src/mobility.cpp:166-190 ([OCV] solvePnP SOLVEPNP_IPPE_SQUARE = undistortPoints + IPPE),
 *   camera -> world position /root/reference/executable/main.cpp:183-192, camera constants main.cpp:7-19. */
typedef struct {
    double camera_matrix[9];   /* row-major 3x3 */
    double dist[5];            /* k1 k2 p1 p2 k3 */
    double gripper2camera[16]; /* row-major 4x4, main.cpp:15-19 */
    float  square_w, square_h; /* main.cpp:184 {27, 27} */
} orc_pnp_config;
void orc_default_pnp_config(orc_pnp_config* c);
int  orc_solve_pnp(const float vertices[4][2], const orc_pnp_config* cfg, double rvec[3], double tvec[3]);
void orc_armour_position(const double tvec[3], const double base2gripper[16], const double gripper2camera[16], double pos[3]);
void orc_locate_armours(const orc_armour* armours, int n, const orc_pnp_config* cfg, const double base2gripper[16] /* NULL = I */,
                        double* rvecs, double* tvecs, double* positions);

/* "Next" row SURVEY 8f-4 (rmcv_oracle_track.c): the observable part of the tracker, core.cpp:124-162 */
void orc_max_iou(const orc_armour* self, const orc_armour* list, int n, int32_t* index, float* iou);
void orc_identity_max(const int32_t* ids /* ascending */, const int32_t* counts, int n, int32_t* max_id, double* prob);

/* tracker state, layout identical to rmcv_track (include/rmcv_abi.h) */
#define ORC_TRACK_IDS 32
typedef struct {
    orc_armour armour;
    int64_t timestamp;
    int32_t lost_count, identity;
    double  position[3];
    int32_t initialized, n_ids;
    int32_t ids[ORC_TRACK_IDS], counts[ORC_TRACK_IDS];
    double  measurement[6], state_pre[6], state_post[6];
    double  transition[36], measurement_matrix[36], process_noise_cov[36], measurement_noise_cov[36];
    double  error_cov_pre[36], error_cov_post[36], gain[36];
} orc_track;
void orc_track_init(orc_track* t, const orc_armour* a, int32_t identity, int64_t timestamp, const double position[3]);
void orc_track_reset(orc_track* t, double process_noise, double measurement_noise, double error);
int  orc_track_update(orc_track* t, const orc_track* observation, double tick_frequency);
int  orc_track_predict(orc_track* t, int64_t new_timestamp, double tick_frequency);
int  orc_track_step(orc_track* tracking, int32_t* n_tracking, int cap, orc_track* observations, int32_t* n_obs, double tick_frequency);

#ifdef __cplusplus
}
#endif
#endif
