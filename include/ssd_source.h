/*
 * ssd_source.h — C ABI of libssd_source.so: a synthetic frame source standing in for the camera.
 *
 * Replaces, for the driver, the tests and bench.py (there is no camera and no librealsense on the GPU box):
 *   Camera::waitForFrames()          camera.cpp:46-49   — the next frame
 *   rs2::pointcloud::calculate()     pointcloud.cpp:138 — depth frame -> W*H float xyz vertices
 * Synthetic "L515-shaped" clouds of a staircase (SURVEY.md section 8(d) recipe), bit-identical on host and device.
 * The product library (libssd_hip.so, include/ssd_hip.h) neither links nor needs this one.
 * All functions return 0 or a negative SSD_E_* code (include/ssd_hip.h).
 */
#ifndef SSD_SOURCE_H_
#define SSD_SOURCE_H_

#include "ssd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
  int32_t width, height;
  double fx, fy, cx, cy;            /* pinhole intrinsics, pixels */
  double cam_height;                /* camera centre above the ground plane, metres */
  double axis_right[3];             /* camera x axis in scene coordinates (x right, y forward, z up) */
  double axis_down[3];              /* camera y axis */
  double axis_fwd[3];               /* camera z axis (optical axis) */
  int32_t n_steps;
  double first_riser_y;             /* pivot (0, first_riser_y): the first riser passes through it */
  double tread, rise, stair_width, landing;
  double yaw_cos, yaw_sin;          /* stairs rotated about the vertical through the pivot */
  double sigma;                     /* depth noise standard deviation, metres */
  double outlier_frac, outlier_min, outlier_max;  /* fraction of pixels replaced by uniform random depth */
  double invalid_frac;              /* fraction of pixels reported invalid (0,0,0) */
  double max_range;                 /* hits beyond this camera depth are invalid */
  uint64_t seed;
} ssd_scene;

/* One frame per scene. Host version writes nframes*W*H*3 floats to xyz; device version writes to
 * device memory (frame i at d_xyz + i*frame_stride_bytes) on `stream`. Bit-identical outputs. */
int ssd_synth_generate_host(const ssd_scene *scenes, int nframes, float *xyz);
int ssd_synth_generate_device(const ssd_scene *scenes, int nframes, void *d_xyz, size_t frame_stride_bytes,
                              int device, void *stream);
/* synthetic depth frames of the same scenes (depth quantised to depth_units), host and device, bit-identical */
int ssd_synth_depth_host(const ssd_scene *scenes, int nframes, float depth_units, uint16_t *depth);
int ssd_synth_depth_device(const ssd_scene *scenes, int nframes, float depth_units, void *d_depth, size_t frame_stride_bytes,
                           int device, void *stream);
/* camera coordinates of a scene point (x right, y forward, z up) */
int ssd_synth_scene_to_camera(const ssd_scene *scene, const double scene_xyz[3], double camera_xyz[3]);

/* the scene the driver and the compat stairs::Camera generate (SURVEY.md section 8(d) pose, n_steps steps) */
int ssd_source_default_scene(ssd_scene *scene, int width, int height, int n_steps, uint64_t seed);
/* writes "calibration-triangle" and "calibration-points" into `directory` in the reference's own file formats
 * (calibrationTriangle.cpp:127-146, geometricCalibration.cpp:59-71) for three marks on the ground (external world
 * x, y, z each; z is the marks' height in the external world, i.e. the world_z the steps' heights are measured
 * from) seen from the scene's camera pose: what the reference's `calibrate` would leave behind for this camera */
int ssd_source_write_calibration(const ssd_scene *scene, const double world_marks[9], const char *directory);
const char *ssd_source_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* SSD_SOURCE_H_ */
