/*
 * ssd_oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the per-frame
 * point-cloud path of peter-nebe/stair-step-detector
 * (pointcloud.cpp -> transformation.cpp -> segmentation.cpp -> stairs.cpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (libssd_hip.so) never links, loads or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the stages whose reference sources need
 * third-party libraries absent from this image (librealsense2, OpenCV,
 * Boost.QVM): pointcloud.cpp, segmentation.cpp, transformation.cpp.  The
 * reference holds no tests or golden vectors.
 * PINNED against the real reference, compiled in place into oracle/_ref/
 * (see oracle/Makefile): Stairs::serialize (stairs.cpp) and
 * QuadrilateralTest (quadrilateralTest.cpp).
 */
#ifndef SSD_ORACLE_H_
#define SSD_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDO_MAX_BINS 256
#define SSDO_MAX_PLATEAUS 64
#define SSDO_MAX_SCANS 128
#define SSDO_MAX_EDGE_PTS 256
#define SSDO_MAX_STEPS (SSDO_MAX_PLATEAUS + 1)
#define SSDO_LINE_CAP 16384

/* status bits */
#define SSDO_ST_THROW 1        /* reference would have thrown (QuadrilateralTest) */
#define SSDO_ST_OOB_PIXEL 2    /* reference would have written out of the image (quirk Q5) */
#define SSDO_ST_ASSERT 4       /* a reference assert would have fired */

/* configuration.h:27-52 + stream resolution */
typedef struct
{
  int32_t width, height;
  double x_min, x_max, y_min, y_max, z_min, z_max;
  double height_interval;
  double min_height_above_ground;
  double min_step_depth;
} ssdo_config;

/* transformation.cpp:196-215: the constants of GeometricTransformation */
typedef struct
{
  double a[9];   /* camera -> world rotation, row-major */
  double b[3];   /* translation */
  double r2[4];  /* external-world 2-D rotation, row-major */
  double t2[2];  /* external-world 2-D translation */
  double world_z;
} ssdo_calibration;

/* the four horizontal edges, in this order everywhere */
enum { SSDO_FRONT_LEFT = 0, SSDO_FRONT_RIGHT = 1, SSDO_BACK_LEFT = 2, SSDO_BACK_RIGHT = 3 };

typedef struct
{
  int32_t peak_bin;          /* Plateau::height */
  int32_t bin_lo, bin_hi;    /* [heightMin, heightMax] chosen at pointcloud.cpp:304-316 */
  int32_t n_points;          /* plateauPoints.size() */
  int32_t is_step;           /* peak_bin >= minHeight */
  int32_t outline_found;     /* HorizontalEdgesDetector::detect returned edges */
  int32_t valid;
  int32_t n_scans_right, n_scans_left;
  int32_t scans_right[SSDO_MAX_SCANS][3];  /* x, yFirst, ySecond */
  int32_t scans_left[SSDO_MAX_SCANS][3];
  int32_t n_edge_pts[4];
  int32_t line[4][3];          /* BestLine (a,b,c) per horizontal edge */
  double bounds[4][2][2];      /* [edge][inner,outer][x,y] */
  double base_line[3];
  int32_t vedge_found[2];      /* left, right */
  int32_t n_vpts[2];
  int32_t vpts[2][SSDO_MAX_EDGE_PTS][2];
  int32_t best_pt[2][2];
  double vline[2][3];
  int32_t corner_found[4];
  double quad_img[8];          /* frontLeft, frontRight, backLeft, backRight (x,y) */
  double quad_world[8];
  int32_t n_in_quad;
  double mean_z;
} ssdo_plateau;

typedef struct
{
  int32_t status;
  int32_t n_total, n_nonzero, n_inrange, n_oob;
  int32_t n_bins, min_height, min_img_y_extent;
  double x_to_image, y_to_image, xy_ratio;
  uint32_t hist[SSDO_MAX_BINS];
  int32_t n_peaks_raw, n_peaks;
  int32_t peaks_raw[SSDO_MAX_BINS], peaks[SSDO_MAX_BINS];
  int32_t n_plateaus;
  int32_t ground_ind, first_valid_ind;
  /* ground (calcGround) */
  double ground_quad_world[8];
  int32_t ground_n_in_quad;
  double ground_mean_z;
  int32_t ground_front_valid;
  int32_t ground_n_pts;
  int32_t ground_pts[SSDO_MAX_SCANS][2];
  int32_t ground_line[3];
  double ground_front_img[4];     /* pointLeft, pointRight (x,y) */
  /* results */
  int32_t n_steps;
  double steps_world[SSDO_MAX_STEPS][12]; /* 4 x (x,y,z), camera-dependent world */
  double steps_ext[SSDO_MAX_STEPS][9];    /* height, 4 x (x,y), external world */
  char line[SSDO_LINE_CAP];               /* Stairs::serialize() text */
  ssdo_plateau plateaus[SSDO_MAX_PLATEAUS];
} ssdo_result;

void ssdo_default_config(ssdo_config *cfg, int width, int height);

/* GeometricTransformation(worldPoints, cameraPoints); 0 = ok, <0 = a reference assert would fire */
int ssdo_calibration_from_points(const double world[9], const double cam[9], ssdo_calibration *out);

/* GeometricCalibration::load(): the two calibration text files -> world / camera points (9 doubles each).
 * 0 = both loaded, -1 = triangle missing/invalid, -2 = points missing/invalid (the reference then uses identity) */
int ssdo_calibration_load(const char *triangle_path, const char *points_path, double world[9], double cam[9]);

/*
 * Pointcloud::process on one frame of width*height float xyz (AoS).
 * raw_images / closed_images: optional, n_step_plateaus x H x W bytes each (step
 * plateaus in ascending order); ground_raw / ground_closed: optional H x W bytes.
 * max_images bounds how many step images are written.
 */
int ssdo_process(const ssdo_config *cfg, const ssdo_calibration *cal, const float *xyz,
                 ssdo_result *out,
                 uint8_t *raw_images, uint8_t *closed_images, int max_images,
                 uint8_t *ground_raw, uint8_t *ground_closed);

/* timing entry: same path, no intermediates kept; returns number of steps or <0 */
int ssdo_process_lean(const ssdo_config *cfg, const ssdo_calibration *cal, const float *xyz,
                      double *steps_ext /* [SSDO_MAX_STEPS][9] */, int *status);

/* The same on many host threads at once (ssd_oracle_mt.cpp; bench.py's cpu_baseline_all_cores): thread t, pinned to cpus[t] (NULL:
 * unpinned), copies frames[t % n_distinct] into memory of its own and runs ssdo_process_lean on it `reps` times behind a common
 * start line.  wall_seconds: first start to last end; read_gb_per_s: what the same threads read from their private frames in a
 * plain loop afterwards (the host memory's rate for them); steps_total: steps found, summed.  Returns 0, or <0. */
int ssdo_process_many(const ssdo_config *cfg, const ssdo_calibration *cal, const float *const *frames, int n_distinct,
                      const int *cpus, int n_threads, int reps, double *wall_seconds, double *read_gb_per_s, long long *steps_total);

/* rs2::pointcloud::calculate (pointcloud.cpp:138) for an undistorted depth stream, as librealsense2 2.42.0 computes it
 * in float (third party, absent from the reference tree: restated from src/proc/pointcloud.cpp + rsutil.h; parity unpinned):
 * d = raw * depth_units; x = d * ((u - ppx) / fx); y = d * ((v - ppy) / fy); z = d; raw 0 -> (0,0,0) */
void ssdo_deproject(float fx, float fy, float ppx, float ppy, float depth_units, int width, int height,
                    const uint16_t *depth, float *xyz);

/* EXTENSION — no reference counterpart (the reference leaves the vertical faces as a TODO, pointcloud.cpp:285-294):
 * the CPU statement of the riser evidence that libssd_hip.so gathers with ssd_set_risers (include/ssd_hip.h,
 * DESIGN.md section 7), to check the HIP kernel against.  Runs ssdo_process first. */
typedef struct
{
  int32_t n_points, detected;
  double height_bottom, height_top;
  double left[2], right[2];
  double mean_offset;
} ssdo_riser;
/* returns the number of risers (emitted surfaces - 1, or 0) or <0; out has room for SSDO_MAX_STEPS entries */
int ssdo_risers(const ssdo_config *cfg, const ssdo_calibration *cal, const float *xyz, double tolerance, int min_support,
                ssdo_riser *out);

/* pieces exposed for unit tests */
void ssdo_close3x3(uint8_t *img, int width, int height);
int ssdo_serialize(int n_steps, const double *steps_ext /* n x 9 */, char *buf, int cap);
/* returns 0 and fills inside[n] ; <0 if the reference constructor would throw (code) */
int ssdo_quad_test(const double quad[8], const double *pts_xy, int n, uint8_t *inside);
double ssdo_hypot(double a, double b);
int ssdo_best_line(const int32_t *pts_xy, int n, int32_t line_out[3]);
/* LineCoordinates<T> (types.h:117-163): the line through two points (pq = x1, y1, x2, y2) and det / detx / dety of two lines */
void ssdo_line_d(const double pq[4], double abc[3]);
void ssdo_line_i(const int32_t pq[4], int32_t abc[3]);
void ssdo_line_dets_d(const double l[3], const double o[3], double out[3]);
/* the permutation std::sort leaves (keys compared by distance only, as segmentation.cpp:724): perm[k] = original index
 * of the element at sorted position k */
void ssdo_sort_perm(const double *dist, int n, int32_t *perm);

#ifdef __cplusplus
}
#endif

#endif /* SSD_ORACLE_H_ */
