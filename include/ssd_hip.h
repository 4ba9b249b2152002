/*
 * ssd_hip.h — C ABI of libssd_hip.so: the MI355X (gfx950) implementation of the
 * per-frame point-cloud path of peter-nebe/stair-step-detector.
 *
 * Drop-in boundary (reference file:line each entry point replaces; paths are
 * relative to the reference tree):
 *
 *   ssd_calibration_from_points  GeometricTransformation::GeometricTransformation(worldPoints, cameraPoints)
 *                                transformation.cpp:196-215 (+ :108-157, :94-106)
 *   ssd_create / ssd_destroy     Pointcloud::Pointcloud(window, trans)   pointcloud.cpp:602-606, pointcloud.h:32-42
 *                                + the compile-time Configuration          configuration.h:27-52, pointcloud.cpp:99-106
 *   ssd_process_host             Pointcloud::process(const Camera::DepthFrame&)   pointcloud.cpp:608-626
 *                                (the frame's xyz vertices = rs2::pointcloud::calculate output, pointcloud.cpp:138)
 *   ssd_enqueue / ssd_fetch      the same, for frames already resident in device memory, asynchronous
 *   ssd_serialize                Stairs::serialize()                     stairs.cpp:55-70 (byte-exact text line)
 *   ssd_get_debug                integer intermediates for parity tests (hist, peaks, images, scans, lines)
 *
 * Not in this library: the synthetic frame source that stands in for the camera (include/ssd_source.h,
 * libssd_source.so) and the hooks that let tests run pieces of the kernels in isolation (include/ssd_testhooks.h,
 * libssd_testhooks.so).
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return
 * 0 on success or a negative SSD_E_* code; nothing throws across the boundary.
 * A handle is bound to one device and is not thread-safe (one host thread per GPU).
 * Streams: see ssd_config::batches_in_flight.  With one workspace a handle's calls execute on the caller's stream in the
 * order they were made; if a call names another stream than the previous one, the library orders it behind the previous
 * call with an event — correct, but the two do not overlap.  With several workspaces the batches run on streams of the
 * handle's own.  ssd_process_host / ssd_process_depth_host always use two non-blocking streams of the handle's own (copy and
 * compute) and return when the results are on the host; they do not synchronise with the legacy default stream.
 * Configuration limits: max(|z_min|, |z_max|) < 2048 m and max|z| * width * height < 2^23 (the mean height of a step
 * is accumulated in 2^-40 m fixed point), 3..SSD_MAX_BINS histogram bins, width <= 8192, height <= 8064.
 */
#ifndef SSD_HIP_H_
#define SSD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSD_MAX_BINS 128          /* histogram bins the kernels support (reference default: 121) */
#define SSD_MAX_PLATEAUS 32       /* filtered histogram peaks per frame */
#define SSD_MAX_STEP_IMAGES 16    /* plateaus at or above minHeight per frame (each owns a bit image) */
#define SSD_MAX_STEPS (SSD_MAX_STEP_IMAGES + 1)
#define SSD_MAX_PLANES 24         /* single-pass batches: bit images of candidate height bins a frame can have */
#define SSD_POOL_PLANES_PER_FRAME 10   /* ... and how many the workspace holds per frame of max_frames_per_batch (a pool: frames draw what they need) */
#define SSD_MAX_SCANS 128
#define SSD_MAX_EDGE_PTS 256
#define SSD_LINE_CAP 4096
#define SSD_BATCHES_IN_FLIGHT_THROUGHPUT 3   /* ssd_config::batches_in_flight of a caller that enqueues ahead of its fetches */

/* error codes */
#define SSD_OK 0
#define SSD_E_ARG (-1)
#define SSD_E_HIP (-2)
#define SSD_E_NOMEM (-3)
#define SSD_E_NODEVICE (-4)
#define SSD_E_CAP (-5)

/* per-frame status bits (ssd_frame_result.status) */
#define SSD_ST_THROW 1         /* reference would have thrown std::invalid_argument (quadrilateralTest.cpp:283-372): no line */
#define SSD_ST_OOB_PIXEL 2     /* a point fell on pixel column W / row H (reference quirk Q5); dropped */
#define SSD_ST_ASSERT 4        /* a reference assert would have fired (segmentation.cpp:549-550, :254) */
#define SSD_ST_OVERFLOW 8      /* more plateaus than SSD_MAX_PLATEAUS / max_step_plateaus: frame truncated */

/* configuration.h:27-52 plus stream resolution and workspace sizing */
typedef struct
{
  int32_t width, height;
  double x_min, x_max, y_min, y_max, z_min, z_max;   /* Configuration::MeasuringRange */
  double height_interval;                            /* 0.01 */
  double min_height_above_ground;                    /* 0.05 */
  double min_step_depth;                             /* 0.1 */
  int32_t max_frames_per_batch;                      /* a workspace is sized for this many frames per call (ssd_workspace_bytes: about 1.75 MB per XGA
                                                        frame; 4.1 MB from 64 XGA frames' worth of points per batch on - such batches, as
                                                        vertices, run K1 and the step plateaus' raster as ONE pass over the input, with
                                                        SSD_MAX_PLANES more bit images per frame; results are the same either way) */
  int32_t max_step_plateaus;                         /* <= SSD_MAX_STEP_IMAGES */
  /* Workspaces of the handle = batches it keeps in flight (1..8; 0 = 1).
   * 1 (the default): every call runs on the caller's stream, strictly in stream order — enqueue, then refill the same frames
   *      on the same stream (or, on the NULL stream, with a plain hipMemcpy) is ordered, nothing to read further.
   * > 1 (opt-in; SSD_BATCHES_IN_FLIGHT_THROUGHPUT = 3 is what bench.py and detect-stairs-amd ask for): successive ssd_enqueue
   *      calls take the workspaces in turn, each on a stream of the handle's own, so that the launches of one batch fill the
   *      gaps the one-block-per-frame kernels of the others leave.  XGA, frames/s with 1 / 2 / 3 / 4 / 6 in flight
   *      (profiles/r03_depths.json): 1024 frames per call 272 k / 290 k / 301 k / 294 k / 307 k, 256: 237 k / 286 k / 306 k /
   *      295 k / 314 k, 64: 188 k / 248 k / 270 k / 251 k / 284 k, 16: 94 k / 165 k / 214 k / 170 k / 228 k (3 and 6 sit better
   *      than 4 and 5) — provided the caller enqueues ahead of its fetches.  Memory = batches_in_flight x 1.75 MB per XGA frame
   *      of max_frames_per_batch (1024 frames, 3 in flight: 5.4 GB beside 9.7 GB of frames; 4.1 MB per frame and 12.6 GB with the
   *      planes of the single pass, ssd_set_single_pass).  Stream contract then: a batch
   *      starts behind the work `stream` holds at the time of the call, but work put on `stream` afterwards is NOT ordered
   *      behind the batch — its frames must stay untouched until its results were fetched, or until a stream was made to wait
   *      for it with ssd_stream_wait.  ssd_process_host / ssd_process_depth_host cut a host batch into slices of at most 32
   *      frames, copied through two staging buffers on a copy stream of the handle's own; with one workspace the slices'
   *      kernels follow each other on the handle's compute stream, with several they take the workspaces in turn like any other
   *      batches and overlap (a staging buffer is refilled only after the kernels of the slice that read it have finished on
   *      THEIR stream).  Both calls return when every slice's results are in `results`; they use no stream of the caller's. */
  int32_t batches_in_flight;
} ssd_config;

/* the constants of GeometricTransformation (transformation.h:102-126) */
typedef struct
{
  double a[9];    /* camera -> camera-dependent world, row-major rotation */
  double b[3];    /* translation */
  double r2[4];   /* ToExternalWorld 2-D rotation, row-major */
  double t2[2];   /* ToExternalWorld 2-D translation */
  double world_z; /* ToExternalWorld::_worldZ */
} ssd_calibration;

/* Stairs::StairStep (stairs.h:32-36), external world coordinates */
typedef struct
{
  double height;    /* mean z of the plateau's points inside its quadrilateral + world_z (pointcloud.cpp:574-581, transformation.cpp:209-211):
                       summed in 2^-40 m fixed point, order-independent — within 3e-14 m of, not bit-equal to, the reference's running double sum */
  double quad[8];   /* quadrilateral[0..3] as x,y pairs: front-left, front-right, back-left, back-right */
} ssd_step;

typedef struct
{
  int32_t n_steps;   /* Stairs::stairSteps.size() */
  int32_t status;    /* SSD_ST_* bits */
  ssd_step steps[SSD_MAX_STEPS];
} ssd_frame_result;

typedef struct ssd_handle ssd_handle;

/* ---- configuration / calibration (host only, no GPU needed) ------------- */
int ssd_default_config(ssd_config *cfg, int width, int height);
int ssd_calibration_from_points(const double world_points[9], const double camera_points[9], ssd_calibration *out);
int ssd_calibration_identity(ssd_calibration *out);   /* GeometricTransformation() default, transformation.h:51-55 */
/* GeometricCalibration::load() (geometricCalibration.cpp:185-203): reads the two text files the calibration
 * step leaves in the working directory — "calibration-triangle" (calibrationTriangle.cpp:97-125: header line
 * `calibration triangle`, `xN = v, yN = v, zN = v` for N = 1..3, `lowerQuadrant = left|right`) and
 * "calibration-points" (geometricCalibration.cpp:73-98: header line `calibration points`, 10 rows of
 * `x, y, z; x, y, z; x, y, z`) — averages the 10 point sets and builds the transformation.
 * Returns SSD_OK and *loaded = 1; if a file is missing or invalid the reference logs an error and carries on
 * with the identity transformation (:199-202): *loaded = 0, identity in *out, still SSD_OK.
 * world_points / camera_points (9 doubles each, may be NULL) receive what was read. */
int ssd_calibration_load(const char *triangle_path, const char *points_path, ssd_calibration *out, int *loaded,
                         double *world_points, double *camera_points);

/* ---- lifetime ------------------------------------------------------------ */
int ssd_create(const ssd_config *cfg, const ssd_calibration *cal, int device, ssd_handle **out);
int ssd_destroy(ssd_handle *h);
const char *ssd_last_error(void);
size_t ssd_workspace_bytes(const ssd_handle *h);
/* Batches of 64 XGA frames' worth of points and more, given as vertices, run K1 and the raster of the step plateaus as ONE pass over
 * the input ("The single pass"): about 10 % more frames/s for 2.4 MB more per XGA frame and workspace
 * (SSD_MAX_PLANES bit images per frame), the results the same bit for bit.  enable = 0 gives that memory back and keeps the
 * handle on two passes; enable != 0 (the default of a handle whose max_frames_per_batch qualifies) takes it again.  Waits for the
 * handle's batches in flight.  No reference counterpart (a deployment knob).
 * The planes are the last thing ssd_create allocates and the one thing it can do without: when they do not fit - or would cross
 * SSD_MAX_PLANE_BYTES, an environment variable that bounds what a handle may take for the planes of all its workspaces together
 * (a GPU shared with other tenants) - the handle is created on two passes, ssd_create returns SSD_OK and ssd_last_error() says why;
 * ssd_set_single_pass(h, 1) under the same shortage fails with SSD_E_NOMEM and leaves the handle whole, on two passes. */
int ssd_set_single_pass(ssd_handle *h, int enable);

/* ---- processing ------------------------------------------------------------
 * A frame is width*height points, AoS float x,y,z (rs2::vertex layout), row-major, invalid = (0,0,0).
 * ssd_process_host:   frames in host memory, contiguous; fills results[n].  Double-buffered: the batch is cut into slices
 *                     of up to 32 frames, the host-to-device copy of a slice overlaps the kernels of the one before.
 *                     Memory from ssd_host_alloc (pinned) is copied by DMA directly; pageable memory goes through the
 *                     runtime's staging.  PCIe-bound either way (9.4 MB per XGA frame, 1.6 MB as 16-bit depth).
 * ssd_enqueue:        frames already in device memory (frame i at d_xyz + i*frame_stride_bytes); enqueues the
 *                     whole pipeline on `stream` (a hipStream_t, NULL = default stream) and returns without
 *                     synchronising. nframes <= max_frames_per_batch.
 * ssd_fetch:          waits for the last ssd_enqueue (whose last step is the copy of its results into pinned host
 *                     memory) and hands its results over.
 */
int ssd_process_host(ssd_handle *h, const float *xyz, int nframes, ssd_frame_result *results);
int ssd_enqueue(ssd_handle *h, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *stream);
int ssd_fetch(ssd_handle *h, ssd_frame_result *results, int nframes, void *stream);
/* The results of a batch travel to the host as part of its ssd_enqueue (max(2, batches_in_flight) slots, used in turn), so
 * a caller may keep the GPU busy: ssd_enqueue(batch i+1) first, then ssd_fetch_back(.., back = 1) for batch i — with 3
 * batches in flight: two enqueues ahead, back = 2.  back = 0 is ssd_fetch.  Waits only for that batch.  A slot is reused by
 * the enqueue max(2, batches_in_flight) calls later: fetch before that. */
int ssd_fetch_back(ssd_handle *h, ssd_frame_result *results, int nframes, int back);
/* makes `stream` (a hipStream_t, NULL = default stream) wait for the batch `back` enqueues ago (counted like ssd_fetch_back: the
 * enqueues that produced results; a partial ssd_enqueue_stages run without the last stage is not one), without blocking the host:
 * what a producer that overwrites the batch's frames, or a consumer of device-side state, needs when the handle keeps
 * several batches in flight (with one workspace the caller's stream is already ordered) */
int ssd_stream_wait(ssd_handle *h, int back, void *stream);
/* the number of workspaces the handle was created with (ssd_config::batches_in_flight resolved) */
int ssd_batches_in_flight(const ssd_handle *h);

/* ---- batches overlapped across handles ------------------------------------------------------------------------
 * One handle runs a batch as a chain of dependent launches; its small kernels (one block per frame or image) leave the
 * GPU nearly empty for about 0.1 ms per batch, which matters the more the fewer frames a batch has.  A pipeline owns
 * `depth` handles (each with its own workspace: memory = depth x ssd_workspace_bytes) on `depth` streams of its own and
 * deals the submitted batches out round-robin; results come back in submission order.  At most `depth` batches are
 * unfetched at any time (ssd_pipeline_submit returns SSD_E_CAP otherwise); the frames of a submitted batch must stay
 * untouched until its results were fetched.  XGA, frames/s at depth 1 / 2 / 3 / 4: 64 frames per
 * batch 181 k / 235 k / 248 k / 232 k, 256: 230 k / 272 k / 282 k / 274 k, 1024: 257 k / 284 k / 304 k / 295 k. */
typedef struct ssd_pipeline ssd_pipeline;
int ssd_pipeline_create(const ssd_config *cfg, const ssd_calibration *cal, int device, int depth, ssd_pipeline **out);
int ssd_pipeline_destroy(ssd_pipeline *p);
/* The pipeline's streams are its own (non-blocking): ssd_pipeline_submit expects the frames to be complete in device memory
 * (their producer synchronised); ssd_pipeline_submit_after orders the batch behind the work `producer_stream` holds at the
 * time of the call (use_producer != 0; producer_stream NULL = the default stream). */
int ssd_pipeline_submit(ssd_pipeline *p, const void *d_xyz, size_t frame_stride_bytes, int nframes);
int ssd_pipeline_submit_after(ssd_pipeline *p, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *producer_stream, int use_producer);
/* results of the OLDEST unfetched batch (waits for it); *nframes = its frame count; capacity = length of `results` */
int ssd_pipeline_next(ssd_pipeline *p, ssd_frame_result *results, int capacity, int *nframes);
int ssd_pipeline_pending(const ssd_pipeline *p);        /* batches submitted and not yet fetched */
/* per-stage device times for the pipeline's batches (ssd_set_timing on every handle; events between the launches of a batch do
 * not keep the batches of different handles from overlapping: measured, depth 3 at 1024 frames 302 k frames/s with, 297 k
 * without).  ssd_pipeline_stage_times: the 7 stages (ssd_get_stage_times) of the batch ssd_pipeline_next returned last —
 * valid until that batch's handle is given another one, i.e. call it right after ssd_pipeline_next */
int ssd_pipeline_set_timing(ssd_pipeline *p, int enable);
int ssd_pipeline_stage_times(ssd_pipeline *p, float ms[7]);
const char *ssd_pipeline_last_error(void);

/* ---- 16-bit depth input ---------------------------------------------
 * The step before the path in the reference is rs2::pointcloud::calculate (pointcloud.cpp:138): depth image ->
 * xyz vertices.  librealsense2 2.42.0 (third party, not in the reference tree) computes, in float,
 *     d = raw * depth_units;  x = d * ((u - ppx) / fx);  y = d * ((v - ppy) / fy);  z = d;   raw = 0 -> (0,0,0)
 * (src/proc/pointcloud.cpp pre_compute_x_y_map / get_points, rsutil.h rs2_deproject_pixel_to_point; the L515
 * depth stream has no distortion model).  With intrinsics set, the kernels read the 2-byte depth pixel instead
 * of the 12-byte vertex and deproject on the fly — the same results as deprojecting first, 6x less input. */
typedef struct
{
  float fx, fy, ppx, ppy;       /* rs2_intrinsics of the depth stream */
  float depth_units;            /* metres per raw unit (L515: 0.00025) */
} ssd_intrinsics;

int ssd_set_intrinsics(ssd_handle *h, const ssd_intrinsics *intr);
/* frames of width*height uint16 depth values, row-major; same contracts as ssd_process_host / ssd_enqueue */
int ssd_process_depth_host(ssd_handle *h, const uint16_t *depth, int nframes, ssd_frame_result *results);
int ssd_enqueue_depth(ssd_handle *h, const void *d_depth, size_t frame_stride_bytes, int nframes, void *stream);
/* host restatement of the deprojection (no GPU needed): depth image -> width*height xyz floats */
int ssd_deproject_host(const ssd_intrinsics *intr, int width, int height, const uint16_t *depth, float *xyz);

/* ---- vertical faces -------------------------------------------------------
 * EXTENSION: the reference has no counterpart — it discards the points that belong to no plateau (`remainder`)
 * with a TODO to find the vertical faces in them (pointcloud.cpp:285-294).  Nothing here changes the results above.
 * Between two vertically consecutive emitted surfaces (ground, valid steps; in output order) lies one riser: the
 * vertical rectangle under the FRONT edge of the upper surface, from the lower surface's height to the upper's.
 * Its evidence are the in-range points of no plateau whose height lies strictly between the two surfaces (one
 * height interval away from either), within `tolerance` metres (horizontally) of that front edge's line and
 * between its end points.  DESIGN.md section 7 states the arithmetic; oracle/ holds the same on the CPU. */
#define SSD_MAX_RISERS (SSD_MAX_STEPS - 1)
typedef struct
{
  int32_t n_points;          /* evidence points */
  int32_t detected;          /* n_points >= min_support */
  double height_bottom, height_top;   /* heights of the lower / upper surface (as ssd_step.height) */
  double left[2], right[2];  /* the upper surface's front-left / front-right corner, external world x,y */
  double mean_offset;        /* mean signed horizontal distance of the evidence from the edge line, metres
                                (positive: on the left of the direction front-left -> front-right) */
} ssd_riser;

typedef struct
{
  int32_t n_risers;          /* emitted surfaces - 1, or 0 */
  int32_t reserved;
  ssd_riser risers[SSD_MAX_RISERS];
} ssd_frame_risers;

/* enable != 0: every later ssd_enqueue / ssd_process_* also gathers riser evidence (one more pass over the points
 * of the bins between the surfaces).  tolerance in (0, 1] metres, min_support >= 1. */
int ssd_set_risers(ssd_handle *h, int enable, double tolerance, int min_support);
/* risers of the last enqueue (after ssd_fetch, or instead of it: synchronises `stream`), or of the whole batch of the last
 * ssd_process_host / ssd_process_depth_host call (collected slice by slice); nframes <= what that call processed.  While
 * risers are on, a handle with several workspaces runs its batches one after the other (the riser buffer is single). */
int ssd_fetch_risers(ssd_handle *h, ssd_frame_risers *out, int nframes, void *stream);
/* stage selector for profiling / roofline measurement: runs only the chosen stage(s) of the pipeline */
#define SSD_STAGE_HIST 1       /* K1: transform + crop + bin + histogram */
#define SSD_STAGE_PEAKS 2
#define SSD_STAGE_RASTER 4     /* K2 */
#define SSD_STAGE_OUTLINE 8    /* K3 */
#define SSD_STAGE_QUADS 16
#define SSD_STAGE_INQUAD 32    /* K4 */
#define SSD_STAGE_FINAL 64     /* K5 */
#define SSD_STAGE_ALL 127
int ssd_enqueue_stages(ssd_handle *h, const void *d_xyz, size_t frame_stride_bytes, int nframes, void *stream, int stages);

/* Per-stage device time of the last enqueue, measured with HIP events recorded on the stream the kernels
 * run on: ms[0..6] = hist, peaks, raster, outline, quads, inquad, final.  Enable before enqueueing. */
#define SSD_TIMING_SLOTS 64
int ssd_set_timing(ssd_handle *h, int enable);
int ssd_get_stage_times(ssd_handle *h, float ms[7]);
/* the same for an earlier enqueue: back = 0 is the last, 1 the one before, ... (< SSD_TIMING_SLOTS),
 * so a timed loop can read all its steps after one final synchronisation */
int ssd_get_stage_times_back(ssd_handle *h, int back, float ms[7]);
/* single-pass batches (section "Batches" below) run one kernel in front of the seven stages, k_predict: its time for the same enqueue
 * (0 when the enqueue did not run it) */
int ssd_get_predict_time_back(ssd_handle *h, int back, float *ms);

/* Stairs::serialize(): returns the text length, or SSD_E_CAP. A frame whose status has SSD_ST_THROW
 * serialises to the empty string (the reference process terminates instead of printing). */
int ssd_serialize(const ssd_frame_result *r, char *buf, size_t cap);

/* ---- intermediates for parity tests ---------------------------------------- */
typedef struct
{
  int32_t peak_bin, bin_lo, bin_hi;     /* Plateau::height and the chosen pair */
  int32_t eff_lo, eff_hi;               /* bins that actually feed this plateau (after earlier plateaus took theirs) */
  int32_t n_points;
  int32_t is_step, outline_found, valid;
  int32_t n_scans_right, n_scans_left;
  int32_t scans_right[SSD_MAX_SCANS][3];
  int32_t scans_left[SSD_MAX_SCANS][3];
  int32_t n_edge_pts[4];
  int32_t line[4][3];
  double bounds[4][2][2];
  double base_line[3];
  int32_t vedge_found[2];
  int32_t n_vpts[2];
  int32_t vpts[2][SSD_MAX_EDGE_PTS][2];
  int32_t best_pt[2][2];
  double vline[2][3];
  int32_t corner_found[4];
  double quad_img[8];
  double quad_world[8];
  int32_t quad_err;
  int32_t n_in_quad;
  int64_t sum_z_fix;                    /* sum of round(z * 2^40) over the in-quad points */
  double mean_z;
} ssd_debug_plateau;

typedef struct
{
  int32_t status;
  int32_t n_nonzero, n_inrange, n_oob;
  int32_t n_bins, min_height, min_img_y_extent;
  uint32_t hist[SSD_MAX_BINS];
  int32_t n_peaks;
  int32_t peaks[SSD_MAX_PLATEAUS];
  int32_t n_plateaus, first_step, ground_ind, first_valid_ind;
  double ground_quad_world[8];
  int32_t ground_quad_err;
  int32_t ground_n_in_quad;
  double ground_mean_z;
  int32_t ground_front_valid;
  int32_t ground_n_pts;
  int32_t ground_pts[SSD_MAX_SCANS][2];
  int32_t ground_line[3];
  double ground_front_img[4];
  ssd_debug_plateau plateaus[SSD_MAX_PLATEAUS];
} ssd_debug_frame;

/* Enables debug capture for subsequent enqueues (costs memory and time; off by default).  enable = 1: the records below and
 * the raw / closed images (for which the whole ground image is rastered, not only the pixel strips the bottom scan reads);
 * enable = 2: the records only — the kernels run exactly as without capture and report their intermediates; 0: off. */
int ssd_set_debug(ssd_handle *h, int enable);
/* Copies the debug record of frame `frame` of the last batch. */
int ssd_get_debug(ssd_handle *h, int frame, ssd_debug_frame *out);
/* Raw (pre-close) and closed plateau images of the last batch as H x W bytes (0 / 0xff), like the
 * reference's cv::Mat.  step_slot = index among the step plateaus, or -1 for the ground image.
 * Only valid when capture with images (ssd_set_debug(h, 1)) was on for the batch. */
int ssd_get_debug_image(ssd_handle *h, int frame, int step_slot, int closed, uint8_t *out);

/* pinned (page-locked) host memory for frames handed to ssd_process_host / ssd_process_depth_host.  hipHostMalloc places it on the
 * NUMA node nearest to the calling thread's current device (the runtime's default policy), so on a two-socket node allocate it from
 * the thread that feeds that GPU, after ssd_bind_thread_to_device. */
int ssd_host_alloc(size_t bytes, void **ptr);
int ssd_host_free(void *ptr);

/* plain device-memory helpers so that hosts without a HIP binding can stage frames */
int ssd_device_count(void);
int ssd_device_alloc(int device, size_t bytes, void **d_ptr);
int ssd_device_free(int device, void *d_ptr);
int ssd_device_upload(int device, void *d_dst, const void *src, size_t bytes);
int ssd_device_download(int device, void *dst, const void *d_src, size_t bytes);
int ssd_device_sync(int device);

/* ---- identity and locality of a device (hosts with several GPUs and sockets; SURVEY.md section 8(e): one host thread per GPU) ----
 * pci_bus_id "dddd:bb:dd.f" and uuid identify the physical GPU (a scaling run proves its N devices distinct with them);
 * numa_node / cpu_list = the NUMA node the GPU hangs off and that node's CPUs as sysfs names them
 * (/sys/bus/pci/devices/<id>/numa_node, local_cpulist); -1 / "" where the platform does not say. */
typedef struct
{
  char pci_bus_id[32];
  char uuid[40];
  int32_t numa_node;
  int32_t n_local_cpus;
  char cpu_list[256];
} ssd_device_info;
int ssd_device_info_get(int device, ssd_device_info *out);
/* Binds the CALLING host thread to the CPUs local to `device` (sched_setaffinity on the thread), so that the thread that feeds
 * a GPU — and the pinned staging memory it allocates afterwards, placed by first touch — sit on the GPU's socket: what the
 * host-fed path needs on a two-socket 8-GPU node (frames resident in HBM do not care).  Returns the number of CPUs bound,
 * 0 when the platform names none (affinity left as it was), or a negative SSD_E_* code. */
int ssd_bind_thread_to_device(int device);

#ifdef __cplusplus
}
#endif

#endif /* SSD_HIP_H_ */
