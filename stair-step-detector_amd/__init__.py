"""Host-side Python binding of libssd_hip.so (ctypes over the C ABI in include/ssd_hip.h).

The package name carries the reference's name and therefore a hyphen; import it with
``importlib.import_module("stair-step-detector_amd")``.

Mirrors the reference's interface for the per-frame path:

* ``GeometricTransformation(world_points, camera_points)``  -> reference transformation.h:102-126
* ``Pointcloud(window, trans).process(frame)``              -> reference pointcloud.h:32-42, pointcloud.cpp:608-626
* ``Stairs.serialize()``                                    -> reference stairs.cpp:55-70

There is no CPU implementation here: every compute call goes to the HIP library and fails
loudly (``SsdError``) when the library or a GPU is missing.  The CPU oracle lives under
``oracle/`` and is never imported by this package.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSD_HIP_LIB") or os.path.join(_HERE, "lib", "libssd_hip.so")   # override: sanitizer builds (tools/)

MAX_BINS = 128
MAX_PLATEAUS = 32
MAX_STEP_IMAGES = 16
MAX_PLANES = 24
POOL_PLANES_PER_FRAME = 10


def plane_pool_size(frames, points_per_frame):
    """planes a workspace holds for batches of up to `frames` frames (csrc/ssd_device.h: plane_pool_size)"""
    per = 16 if points_per_frame < 600000 else POOL_PLANES_PER_FRAME
    return max(frames * per + 2 * MAX_PLANES, min(frames, 8) * MAX_PLANES)
MAX_STEPS = MAX_STEP_IMAGES + 1
MAX_SCANS = 128
MAX_EDGE_PTS = 256
LINE_CAP = 4096

ST_THROW, ST_OOB_PIXEL, ST_ASSERT, ST_OVERFLOW = 1, 2, 4, 8
STAGE_HIST, STAGE_PEAKS, STAGE_RASTER, STAGE_OUTLINE, STAGE_QUADS, STAGE_INQUAD, STAGE_FINAL = 1, 2, 4, 8, 16, 32, 64
STAGE_ALL = 127
STAGE_NAMES = ("hist", "peaks", "raster", "outline", "quads", "inquad", "final")
E_NODEVICE = -4
BATCHES_IN_FLIGHT_THROUGHPUT = 3     # SSD_BATCHES_IN_FLIGHT_THROUGHPUT


class SsdError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("x_min", C.c_double), ("x_max", C.c_double), ("y_min", C.c_double), ("y_max", C.c_double),
                ("z_min", C.c_double), ("z_max", C.c_double),
                ("height_interval", C.c_double), ("min_height_above_ground", C.c_double), ("min_step_depth", C.c_double),
                ("max_frames_per_batch", C.c_int32), ("max_step_plateaus", C.c_int32),
                ("batches_in_flight", C.c_int32)]


class Calibration(C.Structure):
    _fields_ = [("a", C.c_double * 9), ("b", C.c_double * 3), ("r2", C.c_double * 4), ("t2", C.c_double * 2),
                ("world_z", C.c_double)]


class Riser(C.Structure):
    """ssd_riser (extension: vertical faces, include/ssd_hip.h)"""
    _fields_ = [("n_points", C.c_int32), ("detected", C.c_int32), ("height_bottom", C.c_double), ("height_top", C.c_double),
                ("left", C.c_double * 2), ("right", C.c_double * 2), ("mean_offset", C.c_double)]


class Step(C.Structure):
    _fields_ = [("height", C.c_double), ("quad", C.c_double * 8)]


class FrameResult(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("status", C.c_int32), ("steps", Step * MAX_STEPS)]


class FrameRisers(C.Structure):
    _fields_ = [("n_risers", C.c_int32), ("reserved", C.c_int32), ("risers", Riser * (MAX_STEPS - 1))]


class DebugPlateau(C.Structure):
    _fields_ = [("peak_bin", C.c_int32), ("bin_lo", C.c_int32), ("bin_hi", C.c_int32),
                ("eff_lo", C.c_int32), ("eff_hi", C.c_int32), ("n_points", C.c_int32),
                ("is_step", C.c_int32), ("outline_found", C.c_int32), ("valid", C.c_int32),
                ("n_scans_right", C.c_int32), ("n_scans_left", C.c_int32),
                ("scans_right", (C.c_int32 * 3) * MAX_SCANS), ("scans_left", (C.c_int32 * 3) * MAX_SCANS),
                ("n_edge_pts", C.c_int32 * 4), ("line", (C.c_int32 * 3) * 4),
                ("bounds", ((C.c_double * 2) * 2) * 4), ("base_line", C.c_double * 3),
                ("vedge_found", C.c_int32 * 2), ("n_vpts", C.c_int32 * 2),
                ("vpts", ((C.c_int32 * 2) * MAX_EDGE_PTS) * 2), ("best_pt", (C.c_int32 * 2) * 2),
                ("vline", (C.c_double * 3) * 2), ("corner_found", C.c_int32 * 4),
                ("quad_img", C.c_double * 8), ("quad_world", C.c_double * 8),
                ("quad_err", C.c_int32), ("n_in_quad", C.c_int32), ("sum_z_fix", C.c_int64), ("mean_z", C.c_double)]


class DebugFrame(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_nonzero", C.c_int32), ("n_inrange", C.c_int32), ("n_oob", C.c_int32),
                ("n_bins", C.c_int32), ("min_height", C.c_int32), ("min_img_y_extent", C.c_int32),
                ("hist", C.c_uint32 * MAX_BINS), ("n_peaks", C.c_int32), ("peaks", C.c_int32 * MAX_PLATEAUS),
                ("n_plateaus", C.c_int32), ("first_step", C.c_int32), ("ground_ind", C.c_int32),
                ("first_valid_ind", C.c_int32),
                ("ground_quad_world", C.c_double * 8), ("ground_quad_err", C.c_int32),
                ("ground_n_in_quad", C.c_int32), ("ground_mean_z", C.c_double),
                ("ground_front_valid", C.c_int32), ("ground_n_pts", C.c_int32),
                ("ground_pts", (C.c_int32 * 2) * MAX_SCANS), ("ground_line", C.c_int32 * 3),
                ("ground_front_img", C.c_double * 4),
                ("plateaus", DebugPlateau * MAX_PLATEAUS)]


class DeviceInfo(C.Structure):
    """ssd_device_info: which physical GPU a device index is, and the CPUs next to it"""
    _fields_ = [("pci_bus_id", C.c_char * 32), ("uuid", C.c_char * 40), ("numa_node", C.c_int32), ("n_local_cpus", C.c_int32),
                ("cpu_list", C.c_char * 256)]


class Intrinsics(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("ppx", C.c_float), ("ppy", C.c_float), ("depth_units", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("cam_height", C.c_double),
                ("axis_right", C.c_double * 3), ("axis_down", C.c_double * 3), ("axis_fwd", C.c_double * 3),
                ("n_steps", C.c_int32),
                ("first_riser_y", C.c_double), ("tread", C.c_double), ("rise", C.c_double),
                ("stair_width", C.c_double), ("landing", C.c_double),
                ("yaw_cos", C.c_double), ("yaw_sin", C.c_double),
                ("sigma", C.c_double),
                ("outlier_frac", C.c_double), ("outlier_min", C.c_double), ("outlier_max", C.c_double),
                ("invalid_frac", C.c_double), ("max_range", C.c_double),
                ("seed", C.c_uint64)]


# libssd_hip.so — the product ABI (include/ssd_hip.h)
EXPORTS = [
    "ssd_default_config", "ssd_calibration_from_points", "ssd_calibration_identity", "ssd_calibration_load",
    "ssd_create", "ssd_destroy", "ssd_last_error", "ssd_workspace_bytes",
    "ssd_process_host", "ssd_enqueue", "ssd_fetch", "ssd_enqueue_stages",
    "ssd_set_intrinsics", "ssd_process_depth_host", "ssd_enqueue_depth", "ssd_deproject_host",
    "ssd_fetch_back", "ssd_stream_wait", "ssd_batches_in_flight", "ssd_set_risers", "ssd_set_single_pass", "ssd_fetch_risers", "ssd_set_timing", "ssd_get_stage_times", "ssd_get_stage_times_back", "ssd_get_predict_time_back", "ssd_serialize",
    "ssd_set_debug", "ssd_get_debug", "ssd_get_debug_image",
    "ssd_device_count", "ssd_device_alloc", "ssd_device_free", "ssd_device_upload", "ssd_device_download",
    "ssd_device_sync", "ssd_host_alloc", "ssd_host_free", "ssd_device_info_get", "ssd_bind_thread_to_device",
    "ssd_pipeline_create", "ssd_pipeline_destroy", "ssd_pipeline_submit", "ssd_pipeline_submit_after", "ssd_pipeline_next", "ssd_pipeline_pending", "ssd_pipeline_set_timing", "ssd_pipeline_stage_times",
    "ssd_pipeline_last_error",
]
# libssd_source.so — the frame source standing in for the camera (include/ssd_source.h)
SOURCE_EXPORTS = [
    "ssd_synth_generate_host", "ssd_synth_generate_device", "ssd_synth_depth_host", "ssd_synth_depth_device",
    "ssd_synth_scene_to_camera", "ssd_source_default_scene", "ssd_source_write_calibration", "ssd_source_last_error",
]
# libssd_testhooks.so — test infrastructure (include/ssd_testhooks.h)
HOOK_EXPORTS = [
    "ssd_test_hypot_host", "ssd_test_hypot_device", "ssd_test_frame_state", "ssd_test_ground_image", "ssd_test_line_host", "ssd_test_intersect_host", "ssd_test_quad_device", "ssd_test_quad_host", "ssd_test_closing_host", "ssd_test_best_line_host", "ssd_test_grid_boxes_device", "ssd_test_sort_host",
    "ssd_test_sort_device", "ssd_test_stream_read", "ssd_test_empty_quadrilateral", "ssd_test_single_pass", "ssd_test_plane_pool", "ssd_test_single_pass_stats", "ssd_test_single_pass_frame", "ssd_test_single_pass_sample", "ssd_test_predict_table_host", "ssd_test_prexy_host", "ssd_test_prez_host", "ssd_test_quad_edges_host", "ssd_test_quad_edges_device", "ssd_test_record_offset", "ssd_test_record_realloc", "ssd_test_record_realloc_sized", "ssd_test_record_release", "ssd_testhooks_last_error",
]
SOURCE_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libssd_source.so")
HOOKS_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libssd_testhooks.so")

_lib = None
_source_lib = None
_hooks_lib = None


def lib():
    """Loads libssd_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsdError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or make -C stair-step-detector_amd/csrc); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.ssd_last_error.restype = C.c_char_p
    L.ssd_workspace_bytes.restype = sz
    L.ssd_workspace_bytes.argtypes = [vp]
    L.ssd_default_config.argtypes = [C.POINTER(Config), i32, i32]
    L.ssd_calibration_from_points.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Calibration)]
    L.ssd_calibration_identity.argtypes = [C.POINTER(Calibration)]
    L.ssd_calibration_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Calibration), C.POINTER(C.c_int),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ssd_create.argtypes = [C.POINTER(Config), C.POINTER(Calibration), i32, C.POINTER(vp)]
    L.ssd_destroy.argtypes = [vp]
    L.ssd_process_host.argtypes = [vp, vp, i32, C.POINTER(FrameResult)]
    L.ssd_enqueue.argtypes = [vp, vp, sz, i32, vp]
    L.ssd_enqueue_stages.argtypes = [vp, vp, sz, i32, vp, i32]
    L.ssd_fetch.argtypes = [vp, C.POINTER(FrameResult), i32, vp]
    L.ssd_set_intrinsics.argtypes = [vp, C.POINTER(Intrinsics)]
    L.ssd_process_depth_host.argtypes = [vp, vp, i32, C.POINTER(FrameResult)]
    L.ssd_enqueue_depth.argtypes = [vp, vp, sz, i32, vp]
    L.ssd_deproject_host.argtypes = [C.POINTER(Intrinsics), i32, i32, vp, vp]
    L.ssd_set_timing.argtypes = [vp, i32]
    L.ssd_fetch_back.argtypes = [vp, C.POINTER(FrameResult), i32, i32]
    L.ssd_stream_wait.argtypes = [vp, i32, vp]
    L.ssd_batches_in_flight.argtypes = [vp]
    L.ssd_set_risers.argtypes = [vp, i32, C.c_double, i32]
    L.ssd_set_single_pass.argtypes = [vp, i32]
    L.ssd_fetch_risers.argtypes = [vp, C.POINTER(FrameRisers), i32, vp]
    L.ssd_get_stage_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.ssd_get_stage_times_back.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.ssd_get_predict_time_back.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.ssd_serialize.argtypes = [C.POINTER(FrameResult), C.c_char_p, sz]
    L.ssd_set_debug.argtypes = [vp, i32]
    L.ssd_get_debug.argtypes = [vp, i32, C.POINTER(DebugFrame)]
    L.ssd_get_debug_image.argtypes = [vp, i32, i32, i32, vp]
    L.ssd_device_alloc.argtypes = [i32, sz, C.POINTER(vp)]
    L.ssd_device_free.argtypes = [i32, vp]
    L.ssd_device_upload.argtypes = [i32, vp, vp, sz]
    L.ssd_device_download.argtypes = [i32, vp, vp, sz]
    L.ssd_device_sync.argtypes = [i32]
    L.ssd_host_alloc.argtypes = [sz, C.POINTER(vp)]
    L.ssd_device_info_get.argtypes = [i32, C.POINTER(DeviceInfo)]
    L.ssd_bind_thread_to_device.argtypes = [i32]
    L.ssd_host_free.argtypes = [vp]
    L.ssd_pipeline_create.argtypes = [C.POINTER(Config), C.POINTER(Calibration), i32, i32, C.POINTER(vp)]
    L.ssd_pipeline_destroy.argtypes = [vp]
    L.ssd_pipeline_submit.argtypes = [vp, vp, sz, i32]
    L.ssd_pipeline_submit_after.argtypes = [vp, vp, sz, i32, vp, i32]
    L.ssd_pipeline_next.argtypes = [vp, C.POINTER(FrameResult), i32, C.POINTER(i32)]
    L.ssd_pipeline_pending.argtypes = [vp]
    L.ssd_pipeline_set_timing.argtypes = [vp, i32]
    L.ssd_pipeline_stage_times.argtypes = [vp, vp]
    L.ssd_pipeline_last_error.restype = C.c_char_p
    _lib = L
    return L


def source_lib():
    """Loads libssd_source.so (the synthetic frame source: tests, bench, driver); raises if it has not been built."""
    global _source_lib
    if _source_lib is not None:
        return _source_lib
    if not os.path.exists(SOURCE_LIB_PATH):
        raise SsdError("%s is missing: build it with make -C stair-step-detector_amd/csrc" % SOURCE_LIB_PATH)
    L = C.CDLL(SOURCE_LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.ssd_source_last_error.restype = C.c_char_p
    L.ssd_synth_depth_host.argtypes = [C.POINTER(Scene), i32, C.c_float, vp]
    L.ssd_synth_depth_device.argtypes = [C.POINTER(Scene), i32, C.c_float, vp, sz, i32, vp]
    L.ssd_synth_generate_host.argtypes = [C.POINTER(Scene), i32, vp]
    L.ssd_synth_generate_device.argtypes = [C.POINTER(Scene), i32, vp, sz, i32, vp]
    L.ssd_synth_scene_to_camera.argtypes = [C.POINTER(Scene), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ssd_source_default_scene.argtypes = [C.POINTER(Scene), i32, i32, i32, C.c_uint64]
    L.ssd_source_write_calibration.argtypes = [C.POINTER(Scene), C.POINTER(C.c_double), C.c_char_p]
    _source_lib = L
    return L


def hooks_lib():
    """Loads libssd_testhooks.so (test infrastructure, not part of the product ABI)."""
    global _hooks_lib
    if _hooks_lib is not None:
        return _hooks_lib
    if not os.path.exists(HOOKS_LIB_PATH):
        raise SsdError("%s is missing: build it with make -C stair-step-detector_amd/csrc" % HOOKS_LIB_PATH)
    lib()                                           # the hooks take handles of the product library
    L = C.CDLL(HOOKS_LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    L.ssd_testhooks_last_error.restype = C.c_char_p
    L.ssd_test_hypot_host.restype = C.c_double
    L.ssd_test_hypot_host.argtypes = [C.c_double, C.c_double]
    L.ssd_test_hypot_device.argtypes = [i32, vp, vp, vp, i32]
    L.ssd_test_closing_host.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32]
    L.ssd_test_closing_host.restype = i32
    L.ssd_test_best_line_host.argtypes = [vp, i32, i32, vp]
    L.ssd_test_best_line_host.restype = i32
    L.ssd_test_quad_host.argtypes = [vp, vp, i32, vp, vp]
    L.ssd_test_quad_host.restype = i32
    L.ssd_test_frame_state.argtypes = [vp, i32, vp, C.c_size_t, vp]
    L.ssd_test_frame_state.restype = C.c_longlong
    L.ssd_test_ground_image.argtypes = [vp, i32, vp]
    L.ssd_test_empty_quadrilateral.argtypes = [vp, i32, i32]
    L.ssd_test_single_pass.argtypes = [vp, i32, i32]
    L.ssd_test_plane_pool.argtypes = [vp, i32]
    L.ssd_test_single_pass_stats.argtypes = [vp, i32, i32, vp]
    L.ssd_test_single_pass_frame.argtypes = [vp, i32, vp, vp]
    L.ssd_test_single_pass_sample.argtypes = [vp, i32, vp]
    L.ssd_test_predict_table_host.argtypes = [vp, i32, i32, i32, vp]
    L.ssd_test_prexy_host.argtypes = [vp, vp, vp, vp]
    L.ssd_test_prez_host.argtypes = [vp, vp, vp, C.c_double, i32, i32, vp]
    L.ssd_test_quad_edges_host.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.ssd_test_quad_edges_host.restype = i32
    L.ssd_test_quad_edges_device.argtypes = [i32, vp, i32, vp, vp]
    L.ssd_test_quad_edges_device.restype = i32
    L.ssd_test_record_offset.argtypes = [vp, C.c_size_t]
    L.ssd_test_record_realloc.argtypes = [vp]
    L.ssd_test_record_realloc.restype = C.c_ulonglong
    L.ssd_test_record_realloc_sized.argtypes = [vp, C.c_size_t, C.c_size_t]
    L.ssd_test_record_realloc_sized.restype = C.c_ulonglong
    L.ssd_test_line_host.argtypes = [vp, vp, vp]
    L.ssd_test_intersect_host.argtypes = [vp, vp, vp]
    L.ssd_test_sort_host.argtypes = [vp, i32, vp]
    L.ssd_test_sort_device.argtypes = [i32, vp, i32, vp]
    L.ssd_test_quad_device.argtypes = [i32, vp, vp, i32, vp, C.POINTER(C.c_int)]
    L.ssd_test_grid_boxes_device.argtypes = [i32, vp, C.c_double, C.c_double, C.c_double, C.c_double, vp, i32, vp, C.POINTER(C.c_int)]
    L.ssd_test_stream_read.argtypes = [i32, vp, C.c_size_t, i32, vp, C.POINTER(C.c_float)]
    _hooks_lib = L
    return L


def _check(rc, which="hip"):
    if rc < 0:
        msg = {"hip": lambda: lib().ssd_last_error(), "source": lambda: source_lib().ssd_source_last_error(),
               "hooks": lambda: hooks_lib().ssd_testhooks_last_error()}[which]().decode()
        raise SsdError("libssd_%s error %d: %s" % (which, rc, msg))
    return rc


def predict_table_host(sample, n_bins, min_height, sabotage=0):
    """test hook (host, no GPU): the single pass's bin -> plane table as csrc/ssd_predict.h states it; returns (uint8[MAX_BINS], planes)"""
    a = np.ascontiguousarray(sample, dtype=np.uint32)
    assert a.size == MAX_BINS
    out = np.zeros(MAX_BINS, dtype=np.uint8)
    n = _check(hooks_lib().ssd_test_predict_table_host(a.ctypes.data_as(C.c_void_p), n_bins, min_height, sabotage, out.ctypes.data_as(C.c_void_p)), "hooks")
    return out, n


def device_info(device):
    """{'pci_bus_id', 'uuid', 'numa_node', 'n_local_cpus', 'cpu_list'} of a device index (ssd_device_info_get)"""
    info = DeviceInfo()
    _check(lib().ssd_device_info_get(device, C.byref(info)))
    return {"pci_bus_id": info.pci_bus_id.decode(), "uuid": info.uuid.decode(), "numa_node": int(info.numa_node),
            "n_local_cpus": int(info.n_local_cpus), "cpu_list": info.cpu_list.decode()}


def bind_thread_to_device(device):
    """the calling thread onto the CPUs of the device's NUMA node; returns how many (0 = the platform names none)"""
    n = lib().ssd_bind_thread_to_device(device)
    if n < 0:
        _check(n)
    return n


def device_count():
    return lib().ssd_device_count()


def default_config(width, height, max_frames_per_batch=64, max_step_plateaus=MAX_STEP_IMAGES, batches_in_flight=0):
    """batches_in_flight: workspaces of the handle (ssd_config); 0 = 1 = strict stream order; overlap is opt-in
    (BATCHES_IN_FLIGHT_THROUGHPUT = 3 for callers that enqueue ahead of their fetches and leave the frames alone meanwhile)"""
    cfg = Config()
    _check(lib().ssd_default_config(C.byref(cfg), width, height))
    cfg.max_frames_per_batch = max_frames_per_batch
    cfg.max_step_plateaus = max_step_plateaus
    cfg.batches_in_flight = batches_in_flight
    return cfg


# --------------------------------------------------------------------------- reference-shaped classes
class GeometricTransformation:
    """reference transformation.h:102-126; constructor transformation.cpp:196-215."""

    def __init__(self, world_points=None, camera_points=None):
        self.constants = Calibration()
        if world_points is None:
            _check(lib().ssd_calibration_identity(C.byref(self.constants)))
        else:
            w = (C.c_double * 9)(*np.asarray(world_points, dtype=np.float64).reshape(9))
            c = (C.c_double * 9)(*np.asarray(camera_points, dtype=np.float64).reshape(9))
            _check(lib().ssd_calibration_from_points(w, c, C.byref(self.constants)))


class GeometricCalibration:
    """reference geometricCalibration.h:32-37: the offline half."""

    @staticmethod
    def load(directory="."):
        """GeometricCalibration::load() (geometricCalibration.cpp:185-203) -> (GeometricTransformation, loaded)."""
        t = GeometricTransformation()
        loaded = C.c_int(0)
        w, c = (C.c_double * 9)(), (C.c_double * 9)()
        _check(lib().ssd_calibration_load(os.path.join(directory, "calibration-triangle").encode(),
                                          os.path.join(directory, "calibration-points").encode(),
                                          C.byref(t.constants), C.byref(loaded), w, c))
        t.world_points = np.array(w).reshape(3, 3) if loaded.value else None
        t.camera_points = np.array(c).reshape(3, 3) if loaded.value else None
        return t, bool(loaded.value)


class Stairs:
    """reference stairs.h:30-39."""

    def __init__(self, result):
        self.result = result
        self.status = result.status
        self.stair_steps = [(result.steps[i].height, [(result.steps[i].quad[2 * k], result.steps[i].quad[2 * k + 1])
                                                      for k in range(4)]) for i in range(result.n_steps)]

    def serialize(self):
        buf = C.create_string_buffer(LINE_CAP)
        _check(lib().ssd_serialize(C.byref(self.result), buf, LINE_CAP))
        return buf.value.decode()


class Window:
    """reference window.h: the GL sink; has no effect on results."""

    def __init__(self, name=""):
        self.name = name


class Detector:
    """One handle = one device = one `Pointcloud` of the reference, plus the batch entry points."""

    def __init__(self, cfg, trans, device=0):
        self.cfg = cfg
        self.device = device
        self._h = C.c_void_p()
        cal = trans.constants if isinstance(trans, GeometricTransformation) else trans
        _check(lib().ssd_create(C.byref(cfg), C.byref(cal), device, C.byref(self._h)))
        self.frame_bytes = cfg.width * cfg.height * 12

    def close(self):
        if self._h:
            lib().ssd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return lib().ssd_workspace_bytes(self._h)

    @property
    def batches_in_flight(self):
        """workspaces of the handle = batches it keeps in flight (ssd_config::batches_in_flight resolved)"""
        return lib().ssd_batches_in_flight(self._h)

    def stream_wait(self, back=0, stream=None):
        """makes `stream` wait for the batch `back` enqueues ago (ssd_stream_wait)"""
        _check(lib().ssd_stream_wait(self._h, back, C.c_void_p(stream or 0)))

    def process_host(self, xyz):
        """xyz: float32 array [n, H, W, 3] (or [H, W, 3]) on the host -> list of FrameResult."""
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        n = a.size // (self.cfg.width * self.cfg.height * 3)
        if n * self.cfg.width * self.cfg.height * 3 != a.size or n < 1:
            raise SsdError("process_host: array does not hold whole frames")
        res = (FrameResult * n)()
        _check(lib().ssd_process_host(self._h, a.ctypes.data_as(C.c_void_p), n, res))
        return list(res)

    def set_intrinsics(self, intr):
        _check(lib().ssd_set_intrinsics(self._h, C.byref(intr)))

    def process_depth_host(self, depth):
        """depth: uint16 array [n, H, W] (or [H, W]) on the host -> list of FrameResult."""
        a = np.ascontiguousarray(depth, dtype=np.uint16)
        n = a.size // (self.cfg.width * self.cfg.height)
        res = (FrameResult * n)()
        _check(lib().ssd_process_depth_host(self._h, a.ctypes.data_as(C.c_void_p), n, res))
        return list(res)

    def enqueue_depth(self, d_ptr, nframes, stride_bytes=None, stream=None):
        _check(lib().ssd_enqueue_depth(self._h, C.c_void_p(d_ptr), stride_bytes or self.cfg.width * self.cfg.height * 2, nframes,
                                       C.c_void_p(stream or 0)))

    def enqueue(self, d_ptr, nframes, stride_bytes=None, stream=None, stages=STAGE_ALL):
        _check(lib().ssd_enqueue_stages(self._h, C.c_void_p(d_ptr), stride_bytes or self.frame_bytes, nframes,
                                        C.c_void_p(stream or 0), stages))

    def fetch(self, nframes, stream=None, back=0):
        """Waits for the last enqueue (back = 1: the one before it, so that the next batch can already be running) and
        returns its results (an indexable ctypes array of FrameResult; the buffer is reused by the next fetch of the
        same size)."""
        if getattr(self, "_res_n", 0) != nframes:
            self._res, self._res_n = (FrameResult * nframes)(), nframes
        _check(lib().ssd_fetch_back(self._h, self._res, nframes, back))
        return self._res

    def fetch_list(self, nframes, stream=None):
        """fetch() as a list of independent FrameResult copies."""
        return [FrameResult.from_buffer_copy(r) for r in self.fetch(nframes, stream)]

    def set_single_pass(self, on=True):
        """ssd_set_single_pass: off = the handle gives the planes' memory back and stays on two passes; on = the default again"""
        _check(lib().ssd_set_single_pass(self._h, 1 if on else 0))

    def set_risers(self, on=True, tolerance=0.03, min_support=200):
        """extension: also gather the evidence of the vertical faces (ssd_set_risers)"""
        _check(lib().ssd_set_risers(self._h, 1 if on else 0, tolerance, min_support))

    def fetch_risers(self, nframes, stream=None):
        """-> list of FrameRisers (independent copies) of the last enqueue / process_host batch"""
        arr = (FrameRisers * nframes)()
        _check(lib().ssd_fetch_risers(self._h, arr, nframes, C.c_void_p(stream or 0)))
        return [FrameRisers.from_buffer_copy(bytes(r)) for r in arr]

    def set_timing(self, on=True):
        _check(lib().ssd_set_timing(self._h, 1 if on else 0))

    def stage_times_ms(self, back=0):
        """Device time per stage of the enqueue `back` calls ago (HIP events on the kernels' stream)."""
        ms = (C.c_float * 7)()
        _check(lib().ssd_get_stage_times_back(self._h, back, ms))
        return dict(zip(STAGE_NAMES, [float(x) for x in ms]))

    def predict_time_ms(self, back=0):
        """Device time of k_predict, the kernel in front of the stages of a single-pass batch (0.0: the enqueue did not run it)."""
        ms = C.c_float(0.0)
        _check(lib().ssd_get_predict_time_back(self._h, back, C.byref(ms)))
        return float(ms.value)

    def set_debug(self, on=True, images=True):
        """debug capture: records + images (the whole ground image is rastered for it), or records only (images=False:
        the kernels run exactly as in production)"""
        _check(lib().ssd_set_debug(self._h, (1 if images else 2) if on else 0))

    def debug(self, frame=0):
        d = DebugFrame()
        _check(lib().ssd_get_debug(self._h, frame, C.byref(d)))
        return d

    def frame_state(self, frame):
        """test hook: raw device state of one frame after the last enqueue -> (bytes, layout dict)"""
        lay = (C.c_longlong * 8)()
        buf = C.create_string_buffer(1 << 16)
        n = _check(hooks_lib().ssd_test_frame_state(self._h, frame, buf, len(buf), lay), "hooks")
        names = ("size", "hist", "lut", "boxes", "plateaus", "quad_tests", "sum_z", "cnt")
        return buf.raw[:n], dict(zip(names, [int(x) for x in lay]))

    def empty_quadrilateral(self, frame, surface):
        """test hook: rewrites the sums of a surface (-1 = ground, else plateau index) as if its quadrilateral had accepted no point"""
        _check(hooks_lib().ssd_test_empty_quadrilateral(self._h, frame, surface), "hooks")

    def single_pass(self, mode, sabotage=0):
        """test hook: the single pass (K1 rasters the step plateaus itself) -1 = as the product decides, 0 = never, 1 = whenever the
        geometry allows; sabotage 1 / 2 = the predictor's planes in the wrong bins / none (every frame must fall back to k_raster)"""
        _check(hooks_lib().ssd_test_single_pass(self._h, mode, sabotage), "hooks")

    def plane_pool(self, planes=-1):
        """test hook: the planes k_predict may hand out per batch (-1: all the workspace holds); returns the pool's size"""
        return _check(hooks_lib().ssd_test_plane_pool(self._h, planes), "hooks")

    def single_pass_stats(self, frames, scan_planes=True):
        """test hook, of the last enqueue: {'ran': it ran the single pass, 'covered': frames whose step plateaus the planes covered,
        'with_steps': frames with step plateaus, 'planes': planes over all frames, 'dirty_words': words of the lane's plane images
        that are not zero (-1 with scan_planes=False: the images are not fetched)}"""
        counts = (C.c_longlong * 4)()
        ran = _check(hooks_lib().ssd_test_single_pass_stats(self._h, frames, 1 if scan_planes else 0, counts), "hooks")
        return dict(ran=bool(ran), covered=int(counts[0]), with_steps=int(counts[1]), planes=int(counts[2]), dirty_words=int(counts[3]))

    def single_pass_frame(self, frame):
        """test hook, one frame of the last enqueue: (plane of each height bin as a uint8 array, 255 = none; planes; covered; step plateaus)"""
        table = (C.c_uint8 * MAX_BINS)()
        info = (C.c_int32 * 3)()
        _check(hooks_lib().ssd_test_single_pass_frame(self._h, frame, table, info), "hooks")
        return np.frombuffer(bytes(table), dtype=np.uint8).copy(), int(info[0]), bool(info[1]), int(info[2])

    def single_pass_sample(self, frame):
        """test hook: the sample histogram k_predict made the frame's table of (uint32[MAX_BINS])"""
        out = np.zeros(MAX_BINS, dtype=np.uint32)
        _check(hooks_lib().ssd_test_single_pass_sample(self._h, frame, out.ctypes.data_as(C.c_void_p)), "hooks")
        return out

    def record_offset(self, offset_bytes):
        """tools hook: where the first workspace's cell records lie inside their allocation"""
        _check(hooks_lib().ssd_test_record_offset(self._h, offset_bytes), "hooks")

    def record_realloc(self, extra_bytes=0, offset_bytes=0):
        """tools hook: a newly allocated array for the first workspace's cell records (optionally inside a larger allocation);
        returns its device address"""
        return int(hooks_lib().ssd_test_record_realloc_sized(self._h, extra_bytes, offset_bytes))

    def ground_image_raw(self, frame):
        """test hook: the ground bit image as it lies in the last enqueue's workspace (height x width bytes)"""
        out = np.empty((self.cfg.height, self.cfg.width), dtype=np.uint8)
        _check(hooks_lib().ssd_test_ground_image(self._h, frame, out.ctypes.data_as(C.c_void_p)), "hooks")
        return out

    def debug_image(self, frame, step_slot, closed):
        out = np.empty((self.cfg.height, self.cfg.width), dtype=np.uint8)
        _check(lib().ssd_get_debug_image(self._h, frame, step_slot, 1 if closed else 0, out.ctypes.data_as(C.c_void_p)))
        return out


class Pipeline:
    """ssd_pipeline_*: `depth` handles on `depth` streams, batches dealt out round-robin, results in submission order."""

    def __init__(self, cfg, trans, device=0, depth=2):
        self.cfg, self.depth = cfg, depth
        self._p = C.c_void_p()
        cal = trans.constants if isinstance(trans, GeometricTransformation) else trans
        rc = lib().ssd_pipeline_create(C.byref(cfg), C.byref(cal), device, depth, C.byref(self._p))
        if rc < 0:
            raise SsdError("ssd_pipeline_create: %d: %s" % (rc, lib().ssd_pipeline_last_error().decode()))
        self._res = (FrameResult * cfg.max_frames_per_batch)()

    def submit(self, d_ptr, nframes, stride_bytes=None, after_stream=False):
        """after_stream: False = the frames are complete; None / an integer hipStream_t = order the batch behind that stream"""
        stride = stride_bytes or self.cfg.width * self.cfg.height * 12
        if after_stream is False:
            rc = lib().ssd_pipeline_submit(self._p, C.c_void_p(d_ptr), stride, nframes)
        else:
            rc = lib().ssd_pipeline_submit_after(self._p, C.c_void_p(d_ptr), stride, nframes, C.c_void_p(after_stream or 0), 1)
        if rc < 0:
            raise SsdError("ssd_pipeline_submit: %d: %s" % (rc, lib().ssd_pipeline_last_error().decode()))

    def pending(self):
        return lib().ssd_pipeline_pending(self._p)

    def next(self, copy=True):
        """-> the oldest unfetched batch's results: a list of independent FrameResult copies, or (copy=False) the frame count
        with the results left in self.results (reused by the next call)"""
        n = C.c_int(0)
        rc = lib().ssd_pipeline_next(self._p, self._res, len(self._res), C.byref(n))
        if rc < 0:
            raise SsdError("ssd_pipeline_next: %d: %s" % (rc, lib().ssd_pipeline_last_error().decode()))
        if not copy:
            return n.value
        return [FrameResult.from_buffer_copy(self._res[i]) for i in range(n.value)]

    @property
    def results(self):
        return self._res

    def set_timing(self, on=True):
        rc = lib().ssd_pipeline_set_timing(self._p, 1 if on else 0)
        if rc < 0:
            raise SsdError("ssd_pipeline_set_timing: %d: %s" % (rc, lib().ssd_pipeline_last_error().decode()))

    def stage_times_ms(self):
        """Device time per stage of the batch next() returned last (call right after next())."""
        ms = (C.c_float * 7)()
        rc = lib().ssd_pipeline_stage_times(self._p, ms)
        if rc < 0:
            raise SsdError("ssd_pipeline_stage_times: %d: %s" % (rc, lib().ssd_pipeline_last_error().decode()))
        return dict(zip(STAGE_NAMES, [float(v) for v in ms]))

    def close(self):
        if self._p:
            lib().ssd_pipeline_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pointcloud:
    """reference pointcloud.h:32-42: ``Pointcloud(window, trans).process(frame)`` prints one line."""

    def __init__(self, window, trans, device=0):
        self._window, self._trans, self._device = window, trans, device
        self._det = None

    def detect(self, frame):
        a = np.asarray(frame, dtype=np.float32)
        h, w = a.shape[0], a.shape[1]
        if self._det is None or (self._det.cfg.width, self._det.cfg.height) != (w, h):
            self._det = Detector(default_config(w, h, max_frames_per_batch=1), self._trans, self._device)
        st = Stairs(self._det.process_host(a)[0])
        if st.status & ST_THROW:
            raise ValueError("Quadrilateral is not usable (reference quadrilateralTest.cpp:283-372 throws)")
        return st

    def process(self, frame):
        print(self.detect(frame).serialize(), flush=True)


# --------------------------------------------------------------------------- synthetic frame source
def stream_read_ms(d_ptr, nbytes, reps=5, device=0, stream=None):
    """Average milliseconds of a plain read stream over nbytes at d_ptr (measurement hook, libssd_testhooks.so)."""
    ms = C.c_float(0.0)
    _check(hooks_lib().ssd_test_stream_read(device, C.c_void_p(d_ptr), nbytes, reps, C.c_void_p(stream), C.byref(ms)), "hooks")
    return float(ms.value)


def line_host(pq):
    """test hook: the kernels' line through two points -> (abc as doubles, abc as int32 from the truncated coordinates)"""
    a = np.ascontiguousarray(pq, dtype=np.float64).reshape(4)
    d, i = np.zeros(3), np.zeros(3, dtype=np.int32)
    _check(hooks_lib().ssd_test_line_host(a.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p)), "hooks")
    return d, i


def intersect_host(l, o):
    """test hook: the kernels' Line<double>::intersection -> (found, x, y)"""
    a, b, xy = np.ascontiguousarray(l, dtype=np.float64), np.ascontiguousarray(o, dtype=np.float64), np.zeros(2)
    rc = _check(hooks_lib().ssd_test_intersect_host(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p)), "hooks")
    return bool(rc), float(xy[0]), float(xy[1])


def sort_perm(dist, device=None):
    """test hook: libstdc++'s std::sort restated (csrc/ssd_sort.h), on the host (device=None) or on a GPU"""
    d = np.ascontiguousarray(dist, dtype=np.float64)
    perm = np.zeros(len(d), dtype=np.int32)
    if device is None:
        _check(hooks_lib().ssd_test_sort_host(d.ctypes.data_as(C.c_void_p), len(d), perm.ctypes.data_as(C.c_void_p)), "hooks")
    else:
        _check(hooks_lib().ssd_test_sort_device(device, d.ctypes.data_as(C.c_void_p), len(d), perm.ctypes.data_as(C.c_void_p)), "hooks")
    return perm


def quad_test_device(quad, pts, device=0):
    """test hook: the kernels' QuadrilateralTest on one quadrilateral -> (err code, uint8 inside[n])"""
    q = np.ascontiguousarray(quad, dtype=np.float64).reshape(8)
    p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(len(p), dtype=np.uint8)
    err = C.c_int(0)
    _check(hooks_lib().ssd_test_quad_device(device, q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), len(p),
                                            out.ctypes.data_as(C.c_void_p), C.byref(err)), "hooks")
    return err.value, out


def closing_host(img, x0, x_step, y_from=0, band_rows=16, want_closed=True):
    """test hook: the kernels' closing compiled for the host -> (closed uint8 image or None, first[], last[]) for the pixel
    columns x0, x0 + x_step, .. (rows y_from..)"""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = a.shape
    n = (w - 1 - x0) // x_step + 1 if w > x0 else 0
    closed = np.zeros_like(a) if want_closed else None
    first = np.zeros(max(n, 1), dtype=np.int32)
    last = np.zeros(max(n, 1), dtype=np.int32)
    _check(hooks_lib().ssd_test_closing_host(a.ctypes.data_as(C.c_void_p), w, h, x0, x_step, y_from, band_rows,
                                             closed.ctypes.data_as(C.c_void_p) if want_closed else None,
                                             first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), n), "hooks")
    return closed, first[:n], last[:n]


def best_line_host(pts, form):
    """test hook: BestLine with the kernels' residual code compiled for the host -> (a, b, c)"""
    p = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
    out = np.zeros(3, dtype=np.int32)
    _check(hooks_lib().ssd_test_best_line_host(p.ctypes.data_as(C.c_void_p), len(p), form, out.ctypes.data_as(C.c_void_p)), "hooks")
    return tuple(int(v) for v in out)


def quad_test_host(quad, pts):
    """test hook: the same QuadrilateralTest code compiled for the host (no GPU) -> (err code, uint8 inside[n])"""
    q = np.ascontiguousarray(quad, dtype=np.float64).reshape(8)
    p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(len(p), dtype=np.uint8)
    err = C.c_int(0)
    _check(hooks_lib().ssd_test_quad_host(q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), len(p),
                                          out.ctypes.data_as(C.c_void_p), C.byref(err)), "hooks")
    return err.value, out


def grid_boxes_device(quad, x_min, y_min, box_x, box_y, boxes, device=0):
    """test hook: k_inquad's "box of K1's grid wholly inside the quadrilateral" -> (usable, uint8 inside[n])"""
    q = np.ascontiguousarray(quad, dtype=np.float64).reshape(8)
    b = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
    out = np.zeros(len(b), dtype=np.uint8)
    usable = C.c_int(0)
    _check(hooks_lib().ssd_test_grid_boxes_device(device, q.ctypes.data_as(C.c_void_p), x_min, y_min, box_x, box_y,
                                                  b.ctypes.data_as(C.c_void_p), len(b), out.ctypes.data_as(C.c_void_p),
                                                  C.byref(usable)), "hooks")
    return usable.value, out


def make_scene(width, height, n_steps=3, seed=12345, cam_height=1.0, pitch_deg=50.0, roll_deg=0.0,
               first_riser_y=0.45, tread=0.28, rise=0.17, stair_width=0.8, landing=1.0, yaw_deg=0.0,
               sigma=0.001, outlier_frac=0.0, outlier_min=0.3, outlier_max=3.0, invalid_frac=0.0,
               max_range=9.0, hfov_deg=70.0, vfov_deg=55.0):
    """L515-shaped pinhole looking down at a staircase (SURVEY.md section 8(d) recipe)."""
    s = Scene()
    s.width, s.height = width, height
    s.fx = (width / 2.0) / math.tan(math.radians(hfov_deg / 2.0))
    s.fy = (height / 2.0) / math.tan(math.radians(vfov_deg / 2.0))
    s.cx, s.cy = (width - 1) / 2.0, (height - 1) / 2.0
    s.cam_height = cam_height
    p, r = math.radians(pitch_deg), math.radians(roll_deg)
    fwd = np.array([0.0, math.cos(p), -math.sin(p)])
    right0 = np.array([1.0, 0.0, 0.0])
    down0 = np.cross(fwd, right0)
    right = math.cos(r) * right0 + math.sin(r) * down0
    down = np.cross(fwd, right)
    s.axis_right[:] = list(right)
    s.axis_down[:] = list(down)
    s.axis_fwd[:] = list(fwd)
    s.n_steps = n_steps
    s.first_riser_y, s.tread, s.rise, s.stair_width, s.landing = first_riser_y, tread, rise, stair_width, landing
    s.yaw_cos, s.yaw_sin = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    s.sigma = sigma
    s.outlier_frac, s.outlier_min, s.outlier_max = outlier_frac, outlier_min, outlier_max
    s.invalid_frac, s.max_range = invalid_frac, max_range
    s.seed = seed
    return s


def scene_array(scenes):
    arr = (Scene * len(scenes))()
    for i, s in enumerate(scenes):
        C.memmove(C.byref(arr[i]), C.byref(s), C.sizeof(Scene))
    return arr


def synth_host(scenes):
    """-> float32 [n, H, W, 3]; runs on the host, no GPU needed; bit-identical to the device generator."""
    arr = scene_array(scenes)
    h, w = scenes[0].height, scenes[0].width
    out = np.empty((len(scenes), h, w, 3), dtype=np.float32)
    _check(source_lib().ssd_synth_generate_host(arr, len(scenes), out.ctypes.data_as(C.c_void_p)), "source")
    return out


def intrinsics_for_scene(scene, depth_units=0.00025):
    """rs2_intrinsics of the synthetic camera (L515 depth unit: 0.25 mm)."""
    i = Intrinsics()
    i.fx, i.fy, i.ppx, i.ppy, i.depth_units = scene.fx, scene.fy, scene.cx, scene.cy, depth_units
    return i


def synth_depth_host(scenes, depth_units=0.00025):
    """-> uint16 [n, H, W]: the scenes as 16-bit depth frames; host, bit-identical to the device generator."""
    arr = scene_array(scenes)
    out = np.empty((len(scenes), scenes[0].height, scenes[0].width), dtype=np.uint16)
    _check(source_lib().ssd_synth_depth_host(arr, len(scenes), depth_units, out.ctypes.data_as(C.c_void_p)), "source")
    return out


def synth_depth_device(scenes, d_ptr, depth_units=0.00025, stride_bytes=None, device=0, stream=None):
    arr = scene_array(scenes)
    stride = stride_bytes or scenes[0].width * scenes[0].height * 2
    _check(source_lib().ssd_synth_depth_device(arr, len(scenes), depth_units, C.c_void_p(d_ptr), stride, device, C.c_void_p(stream or 0)), "source")


def deproject_host(intr, depth):
    """rs2::pointcloud::calculate restated (host): uint16 [H, W] -> float32 [H, W, 3]."""
    a = np.ascontiguousarray(depth, dtype=np.uint16)
    out = np.empty(a.shape + (3,), dtype=np.float32)
    _check(lib().ssd_deproject_host(C.byref(intr), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    return out


def synth_device(scenes, d_ptr, stride_bytes=None, device=0, stream=None):
    arr = scene_array(scenes)
    stride = stride_bytes or scenes[0].width * scenes[0].height * 12
    _check(source_lib().ssd_synth_generate_device(arr, len(scenes), C.c_void_p(d_ptr), stride, device, C.c_void_p(stream or 0)), "source")


CALIBRATION_MARKS = ((-0.35, 0.9, 0.0), (0.35, 0.9, 0.0), (0.2, 0.35, 0.0))


def calibration_points(scene, marks=CALIBRATION_MARKS, world_offset=(0.0, 0.0, 0.004)):
    """Three ground marks: (external-world points, camera points) as the calibration files would hold them."""
    world, cam = [], []
    for m in marks:
        p = (C.c_double * 3)(*m)
        o = (C.c_double * 3)()
        _check(source_lib().ssd_synth_scene_to_camera(C.byref(scene), p, o), "source")
        cam.append([o[0], o[1], o[2]])
        world.append([m[0] + world_offset[0], m[1] + world_offset[1], m[2] + world_offset[2]])
    return np.array(world), np.array(cam)


def transformation_for_scene(scene):
    world, cam = calibration_points(scene)
    return GeometricTransformation(world, cam)


def prez_host(x_min, x_max, y_min, y_max, z_min, z_max, a, b, height_interval=0.01, width=1024, height=768):
    """test hook: the constants of K1's single-precision z row / bin and candidate pixel (csrc/ssd_prexy.h: make_pre_z, make_pre_pixel)"""
    rng = (C.c_double * 6)(x_min, x_max, y_min, y_max, z_min, z_max)
    aa = (C.c_double * 9)(*[float(v) for v in np.asarray(a, dtype=np.float64).reshape(9)])
    bb = (C.c_double * 3)(*[float(v) for v in np.asarray(b, dtype=np.float64).reshape(3)])
    out = (C.c_float * 16)()
    _check(hooks_lib().ssd_test_prez_host(rng, aa, bb, float(height_interval), int(width), int(height), out), "hooks")
    o = np.array(list(out), dtype=np.float32)
    return dict(zc=o[:4], z_neg_k=o[4], z_h0=o[5], z_top=o[6], z_check_top=bool(o[7]), f_w=o[8], f_half_w=o[9], f_neg_h=o[10], f_half_h=o[11],
                px_neg_k=o[12], px_h0=o[13], recip=float(o[14]))


def quad_edges_host(quad, x_min, x_max, y_min, y_max, z_min, z_max, a, b, pts_xyz):
    """test hook: k_inquad's single-precision edge tests of one quadrilateral (csrc/ssd_quadtest.h: build_quad_edges) on camera points ->
    dict(err, gx, gy, g2, m, d_k, d_e0, cls = int8[n] (+1 / -1 / 0), world_xy = float64[n, 2], in_range_xy = uint8[n])"""
    q = np.ascontiguousarray(quad, dtype=np.float64).reshape(8)
    rng = (C.c_double * 6)(x_min, x_max, y_min, y_max, z_min, z_max)
    aa = (C.c_double * 9)(*[float(v) for v in np.asarray(a, dtype=np.float64).reshape(9)])
    bb = (C.c_double * 3)(*[float(v) for v in np.asarray(b, dtype=np.float64).reshape(3)])
    p = np.ascontiguousarray(pts_xyz, dtype=np.float32).reshape(-1, 3)
    consts = (C.c_float * 15)()
    cls = np.zeros(len(p), dtype=np.int8)
    wxy = np.zeros((len(p), 2), dtype=np.float64)
    inr = np.zeros(len(p), dtype=np.uint8)
    err = C.c_int(0)
    _check(hooks_lib().ssd_test_quad_edges_host(q.ctypes.data_as(C.c_void_p), rng, aa, bb, p.ctypes.data_as(C.c_void_p), len(p), consts,
                                               cls.ctypes.data_as(C.c_void_p), wxy.ctypes.data_as(C.c_void_p), inr.ctypes.data_as(C.c_void_p),
                                               C.byref(err)), "hooks")
    o = np.array(list(consts), dtype=np.float32)
    return dict(err=err.value, gx=o[0:4], gy=o[4:8], g2=o[8:12], m=o[12], d_k=o[13], d_e0=o[14], cls=cls, world_xy=wxy, in_range_xy=inr)


def quad_edges_device(quads, x_min, x_max, y_min, y_max, device=0):
    """test hook: k_inquad's edge table as the device builds it in k_quads' three steps, for n quadrilaterals -> float32[n, 13] (gx, gy, g2, m)"""
    q = np.ascontiguousarray(quads, dtype=np.float64).reshape(-1, 8)
    rng = (C.c_double * 4)(x_min, x_max, y_min, y_max)
    out = np.zeros((len(q), 13), dtype=np.float32)
    _check(hooks_lib().ssd_test_quad_edges_device(device, q.ctypes.data_as(C.c_void_p), len(q), rng, out.ctypes.data_as(C.c_void_p)), "hooks")
    return out


def prexy_host(x_min, x_max, y_min, y_max, z_min, z_max, a, b):
    """test hook: the constants of K1's single-precision pre-filter (csrc/ssd_prexy.h) for a measuring range and a calibration:
    dict(c = 4 x 2 float32 coefficients, lo, hi, max_input, box_lo, box_hi, check_input)"""
    rng = (C.c_double * 6)(x_min, x_max, y_min, y_max, z_min, z_max)
    aa = (C.c_double * 9)(*[float(v) for v in np.asarray(a, dtype=np.float64).reshape(9)])
    bb = (C.c_double * 3)(*[float(v) for v in np.asarray(b, dtype=np.float64).reshape(3)])
    out = (C.c_float * 14)()
    _check(hooks_lib().ssd_test_prexy_host(rng, aa, bb, out), "hooks")
    o = np.array(list(out), dtype=np.float32)
    return dict(c=o[:8].reshape(4, 2), lo=o[8], hi=o[9], max_input=o[10], box_lo=o[11], box_hi=o[12], check_input=bool(o[13]))


class PinnedArray:
    """A numpy array over page-locked host memory (ssd_host_alloc): ssd_process_host copies it by DMA without staging."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(lib().ssd_host_alloc(self.nbytes, C.byref(p)))
        self.ptr = p.value
        self.array = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().ssd_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """hipMalloc'd bytes through the C ABI (for hosts without torch)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, nbytes
        p = C.c_void_p()
        _check(lib().ssd_device_alloc(device, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, array, offset=0):
        a = np.ascontiguousarray(array)
        _check(lib().ssd_device_upload(self.device, C.c_void_p(self.ptr + offset), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, nbytes, offset=0, dtype=np.uint8):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().ssd_device_download(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ssd_device_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
