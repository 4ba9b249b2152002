/*
 * ssd_testhooks.h — C ABI of libssd_testhooks.so: TEST INFRASTRUCTURE, not part of the product ABI.
 * Lets tests run pieces of the kernels in isolation: std::hypot and std::sort as restated for the device, the
 * kernels' QuadrilateralTest on one quadrilateral, a frame's raw device state.
 */
#ifndef SSD_TESTHOOKS_H_
#define SSD_TESTHOOKS_H_

#include "ssd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hooks: std::hypot as the kernels compute it (glibc 2.35 algorithm restated), host and device */
double ssd_test_hypot_host(double a, double b);
int ssd_test_hypot_device(int device, const double *a, const double *b, double *out, int n);
/* test hooks: std::sort as libstdc++ performs it, restated (csrc/ssd_sort.h) for the tie order of segmentation.cpp:724;
 * perm[k] = original index of the key at sorted position k */
int ssd_test_sort_host(const double *dist, int n, int32_t *perm);
int ssd_test_sort_device(int device, const double *dist, int n, int32_t *perm);
/* test hook: QuadrilateralTest (quadrilateralTest.cpp:275-451) exactly as the kernels build and evaluate it, for one
 * quadrilateral (front-left, front-right, back-left, back-right as x,y) and n points; *err = 0 or the code of the
 * reference's throw (-1..-6), in which case `inside` is left zero */
int ssd_test_quad_device(int device, const double quad[8], const double *pts_xy, int n, uint8_t *inside, int *err);
/* the same code (csrc/ssd_quadtest.h: build_quad_test, the constant cell, quad_test) compiled for the host: runs without a GPU */
int ssd_test_quad_host(const double quad[8], const double *pts_xy, int n, uint8_t *inside, int *err);
/* test hook (host): k_inquad's single-precision edge tests (csrc/ssd_quadtest.h: build_quad_edges) for one quadrilateral over a measuring
 * range (x_min, x_max, y_min, y_max, z_min, z_max) and a calibration, on n camera points (x, y, z floats): consts = gx[4], gy[4], g2[4], m
 * (infinity: switched off for this quadrilateral), dK, dE0; cls[i] = +1 inside for sure / -1 outside for sure / 0 ask the doubles, from the d
 * K1's pre-filter computes; world_xy[2 i ..] = the reference's world x, y of the point, in_range_xy[i] = its x / y range test */
int ssd_test_quad_edges_host(const double quad[8], const double range[6], const double a[9], const double b[3], const float *pts_xyz, int n,
                             float consts[15], int8_t *cls, double *world_xy, uint8_t *in_range_xy, int *err);
/* the same table as the DEVICE builds it in k_quads' three steps (coefficients, the check of the map's nine cells dealt out to lanes, the margin) for n
 * quadrilaterals (n x 8 doubles) over the x / y range (x_min, x_max, y_min, y_max): out = n x 13 floats (gx[4], gy[4], g2[4], m) */
int ssd_test_quad_edges_device(int device, const double *quads, int n, const double range_xy[4], float *out);
/* test hook: BestLine (segmentation.cpp:409-487) over n points (x, y int32 pairs) with the kernels' residual code
 * (csrc/ssd_bestline.h) compiled for the host; form 0 = any list, 1 = keys in passes of four (n <= 128), 2 = one pass (n <= 64) */
int ssd_test_best_line_host(const int32_t *pts_xy, int n, int form, int32_t line[3]);
/* test hook: the kernels' 3x3 closing on bit images (csrc/ssd_closing.h) compiled for the host: closed_out (width x height bytes,
 * may be null) = the closed image computed word by word; first / last [n_cols] = first and last closed row of the pixel columns
 * x0, x0 + x_step, .. by the column-wise form the scans use, over rows [y_from, height) cut into bands of band_rows rows */
int ssd_test_closing_host(const uint8_t *img, int width, int height, int x0, int x_step, int y_from, int band_rows, uint8_t *closed_out,
                          int32_t *first, int32_t *last, int n_cols);
/* test hook: k_inquad's cell classification "this box of K1's grid (x0, x1, y0, y1 in cells of 1 / box_x by 1 / box_y metres
 * from (x_min, y_min); boxes = n x 4 int32) lies wholly inside the quadrilateral" (csrc/ssd_quadtest.h: build_grid_segs,
 * grid_box_inside); *usable = 0 when the shortcut switches itself off for this quadrilateral (all boxes then answer 0) */
int ssd_test_grid_boxes_device(int device, const double quad[8], double x_min, double y_min, double box_x, double box_y,
                               const int32_t *boxes, int n, uint8_t *inside, int *usable);
/* test hook: the raw per-frame device state after the last enqueue (layout private to the library; layout[0..7] =
 * sizeof state, offsets of hist, lut, image boxes, plateau table, quadrilateral tests, sums, counts); returns the
 * number of bytes copied or a negative error */
long long ssd_test_frame_state(ssd_handle *h, int frame, void *out, size_t cap, long long layout[8]);
/* test hook: rewrites the sums of one surface of frame `frame` in the workspace of the last enqueue as k_inquad would have
 * left them had the surface's quadrilateral accepted NO point (surface = -1: the ground, else a plateau index: count of the
 * points outside = all of the plateau's, their sum = the plateau's total).  Between ssd_enqueue_stages(.. up to SSD_STAGE_INQUAD)
 * and ssd_enqueue_stages(SSD_STAGE_FINAL): the case calcAverageZ divides 0.0 by 0 in (pointcloud.cpp:574-581). */
int ssd_test_empty_quadrilateral(ssd_handle *h, int frame, int surface);
/* The single pass (K1 rasters the step plateaus itself into planes of predicted height bins).  mode: -1 = as
 * the product decides (whole pipeline, vertex input, a batch of at least 64 XGA frames' worth of points), 0 = never, 1 = whenever the
 * geometry allows (a handle without planes gets them).  sabotage: 0 = none, 1 = the predictor's planes three bins above the right
 * ones, 2 = no planes at all - either way every frame with steps must come out through k_raster, bit-equal. */
int ssd_test_single_pass(ssd_handle *h, int mode, int sabotage);
/* The planes k_predict may hand out per batch (round 5: a pool per workspace, ssd_device.h plane_pool_size): planes = 0 .. the
 * pool's size lowers it (frames the pool cannot serve come out through k_raster, bit-equal), -1 restores it.  Returns the
 * pool's size. */
int ssd_test_plane_pool(ssd_handle *h, int planes);
/* of the last enqueue (synchronises): returns 1 when it ran the single pass, else 0 (counts zero); counts[0] = frames whose step
 * plateaus the planes covered (k_raster skipped them), [1] = frames with step plateaus, [2] = planes over all frames,
 * [3] = 64-bit words of the lane's plane images that are not zero (the invariant between batches: 0) */
int ssd_test_single_pass_stats(ssd_handle *h, int frames, int scan_planes, long long counts[4]);   /* scan_planes = 0: counts[3] = -1, the images are not fetched */
/* one frame of the last enqueue: the predictor's table (height bin -> plane, 0xff = none; SSD_MAX_BINS entries),
 * info[0] = planes, info[1] = 1 when they covered the frame's step plateaus, info[2] = step plateaus */
int ssd_test_single_pass_frame(ssd_handle *h, int frame, uint8_t *plane_of_bin, int info[3]);
/* the same frame's sample histogram (SSD_MAX_BINS counts): what k_predict made that table of */
int ssd_test_single_pass_sample(ssd_handle *h, int frame, uint32_t *sample);
/* the predictor's table as csrc/ssd_predict.h states it (host; no GPU): sample[SSD_MAX_BINS] -> plane_of_bin[SSD_MAX_BINS], returns the planes */
int ssd_test_predict_table_host(const uint32_t *sample, int n_bins, int min_height, int sabotage, uint8_t *plane_of_bin);
/* The constants of K1's single-precision pre-filter of the x / y range test (csrc/ssd_prexy.h: make_pre_xy) for a measuring range and a
 * calibration, on the host: out[0..7] = the four coefficient pairs (x row, y row), out[8] = lo, out[9] = hi, out[10] = the largest input
 * the bound holds for, out[11] / out[12] = the boxes' two offsets, out[13] = 1 when K1 tests the input's magnitude per point, 0 when
 * make_pre_xy showed that a larger input cannot read "inside".  tests/test_prexy.py checks the bound against double precision. */
int ssd_test_prexy_host(const double range[6], const double a[9], const double b[3], float out[14]);
/* round 6 - the z row and the candidates' pixel in single precision first (make_pre_z / make_pre_pixel, csrc/ssd_prexy.h), on the
 * host: out[0..3] = the z row's coefficients in bins, out[4] = -K, out[5] = 1/2 - E0 (sure iff |fract(t) - 1/2| < out[5] + out[4] * max|input|),
 * out[6] = zTop, out[7] = zCheckTop; out[8..11] = W, W/2, -H, H/2, out[12] / out[13] = the pixel test's -K and 1/2 - E0; out[14] = recip */
int ssd_test_prez_host(const double range[6], const double a[9], const double b[3], double height_interval, int width, int height, float out[16]);
/* tools hook (tools/k1place2.py): places the first workspace's cell records `offset_bytes` (a multiple of 8, within the extra bytes a
 * preceding ssd_test_record_realloc_sized asked for) into their allocation; the records' content is undefined afterwards until the
 * next full enqueue */
int ssd_test_record_offset(ssd_handle *h, size_t offset_bytes);
/* tools hook (tools/k1place2.py): gives the first workspace a NEWLY allocated array for its cell records (the previous ones stay
 * allocated until ssd_test_record_release, so that every call lands somewhere else); returns the device address */
unsigned long long ssd_test_record_realloc(ssd_handle *h);
unsigned long long ssd_test_record_realloc_sized(ssd_handle *h, size_t extra_bytes, size_t offset_bytes);   /* a larger allocation, the records `offset_bytes` into it */
int ssd_test_record_release(void);
/* test hook: the ground bit image of one frame as it lies in the workspace of the last enqueue, as height x width bytes (0 / 0xff).
 * After a full enqueue it is all zero (k_final clears what it read); after ssd_enqueue_stages(.. up to SSD_STAGE_INQUAD) it
 * holds what k_inquad rastered: outside image capture only the pixel strips the bottom scan reads */
int ssd_test_ground_image(ssd_handle *h, int frame, uint8_t *out);
/* test hooks: the kernels' line helpers compiled for the host (no GPU needed): the line through (pq[0], pq[1]) and (pq[2], pq[3]) in
 * doubles and in int32 (the coordinates truncated) = LineCoordinates(p, q), types.h:140-158; and Line<double>::intersection
 * (segmentation.cpp:344-362: LineCoordinates::det / detx / dety and the 60-degree rule): returns 1 + the point, or 0 */
int ssd_test_line_host(const double pq[4], double abc_d[3], int32_t abc_i[3]);
int ssd_test_intersect_host(const double l[3], const double o[3], double xy[2]);

/* measurement hook (bench.py, tools/clockstate.py): average milliseconds of `reps` launches of a plain 16-byte-per-lane read
 * stream over `bytes` bytes at d_ptr (tools/loadbench.hip variant C) on `stream` — what the memory system delivers right now */
int ssd_test_stream_read(int device, const void *d_ptr, size_t bytes, int reps, void *stream, float *ms_avg);

const char *ssd_testhooks_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* SSD_TESTHOOKS_H_ */
