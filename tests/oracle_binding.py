"""ctypes binding of the CPU oracle (oracle/libssd_oracle.so) and of oracle/_ref/libssd_ref.so.

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libssd_oracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libssd_ref.so")

MAX_BINS, MAX_PLATEAUS, MAX_SCANS, MAX_EDGE_PTS = 256, 64, 128, 256
MAX_STEPS = MAX_PLATEAUS + 1
LINE_CAP = 16384
ST_THROW, ST_OOB_PIXEL, ST_ASSERT = 1, 2, 4


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("x_min", C.c_double), ("x_max", C.c_double), ("y_min", C.c_double), ("y_max", C.c_double),
                ("z_min", C.c_double), ("z_max", C.c_double),
                ("height_interval", C.c_double), ("min_height_above_ground", C.c_double), ("min_step_depth", C.c_double)]


class Calibration(C.Structure):
    _fields_ = [("a", C.c_double * 9), ("b", C.c_double * 3), ("r2", C.c_double * 4), ("t2", C.c_double * 2),
                ("world_z", C.c_double)]


class Plateau(C.Structure):
    _fields_ = [("peak_bin", C.c_int32), ("bin_lo", C.c_int32), ("bin_hi", C.c_int32), ("n_points", C.c_int32),
                ("is_step", C.c_int32), ("outline_found", C.c_int32), ("valid", C.c_int32),
                ("n_scans_right", C.c_int32), ("n_scans_left", C.c_int32),
                ("scans_right", (C.c_int32 * 3) * MAX_SCANS), ("scans_left", (C.c_int32 * 3) * MAX_SCANS),
                ("n_edge_pts", C.c_int32 * 4), ("line", (C.c_int32 * 3) * 4),
                ("bounds", ((C.c_double * 2) * 2) * 4), ("base_line", C.c_double * 3),
                ("vedge_found", C.c_int32 * 2), ("n_vpts", C.c_int32 * 2),
                ("vpts", ((C.c_int32 * 2) * MAX_EDGE_PTS) * 2), ("best_pt", (C.c_int32 * 2) * 2),
                ("vline", (C.c_double * 3) * 2), ("corner_found", C.c_int32 * 4),
                ("quad_img", C.c_double * 8), ("quad_world", C.c_double * 8),
                ("n_in_quad", C.c_int32), ("mean_z", C.c_double)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_total", C.c_int32), ("n_nonzero", C.c_int32), ("n_inrange", C.c_int32),
                ("n_oob", C.c_int32), ("n_bins", C.c_int32), ("min_height", C.c_int32), ("min_img_y_extent", C.c_int32),
                ("x_to_image", C.c_double), ("y_to_image", C.c_double), ("xy_ratio", C.c_double),
                ("hist", C.c_uint32 * MAX_BINS), ("n_peaks_raw", C.c_int32), ("n_peaks", C.c_int32),
                ("peaks_raw", C.c_int32 * MAX_BINS), ("peaks", C.c_int32 * MAX_BINS),
                ("n_plateaus", C.c_int32), ("ground_ind", C.c_int32), ("first_valid_ind", C.c_int32),
                ("ground_quad_world", C.c_double * 8), ("ground_n_in_quad", C.c_int32), ("ground_mean_z", C.c_double),
                ("ground_front_valid", C.c_int32), ("ground_n_pts", C.c_int32),
                ("ground_pts", (C.c_int32 * 2) * MAX_SCANS), ("ground_line", C.c_int32 * 3),
                ("ground_front_img", C.c_double * 4),
                ("n_steps", C.c_int32), ("steps_world", (C.c_double * 12) * MAX_STEPS),
                ("steps_ext", (C.c_double * 9) * MAX_STEPS), ("line", C.c_char * LINE_CAP),
                ("plateaus", Plateau * MAX_PLATEAUS)]


def build_oracle():
    """Compiles oracle/ (and oracle/_ref when the reference tree is present). Building the checker is not using it."""
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


class Riser(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("detected", C.c_int32), ("height_bottom", C.c_double), ("height_top", C.c_double),
                ("left", C.c_double * 2), ("right", C.c_double * 2), ("mean_offset", C.c_double)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, i32 = C.c_void_p, C.c_int
        lib.ssdo_default_config.argtypes = [C.POINTER(Config), i32, i32]
        lib.ssdo_calibration_from_points.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Calibration)]
        lib.ssdo_process.argtypes = [C.POINTER(Config), C.POINTER(Calibration), vp, C.POINTER(Result), vp, vp, i32, vp, vp]
        lib.ssdo_process_lean.argtypes = [C.POINTER(Config), C.POINTER(Calibration), vp, vp, C.POINTER(i32)]
        lib.ssdo_process_many.argtypes = [C.POINTER(Config), C.POINTER(Calibration), vp, i32, vp, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        lib.ssdo_close3x3.argtypes = [vp, i32, i32]
        lib.ssdo_serialize.argtypes = [i32, vp, C.c_char_p, i32]
        lib.ssdo_quad_test.argtypes = [vp, vp, i32, vp]
        lib.ssdo_hypot.restype = C.c_double
        lib.ssdo_hypot.argtypes = [C.c_double, C.c_double]
        lib.ssdo_best_line.argtypes = [vp, i32, vp]
        lib.ssdo_calibration_load.argtypes = [C.c_char_p, C.c_char_p, vp, vp]
        lib.ssdo_deproject.argtypes = [C.c_float] * 5 + [i32, i32, vp, vp]
        lib.ssdo_sort_perm.argtypes = [vp, i32, vp]
        lib.ssdo_risers.argtypes = [C.POINTER(Config), C.POINTER(Calibration), vp, C.c_double, i32, C.POINTER(Riser)]
        for f in (lib.ssdo_line_d, lib.ssdo_line_i):
            f.argtypes, f.restype = [vp, vp], None
        lib.ssdo_line_dets_d.argtypes, lib.ssdo_line_dets_d.restype = [vp, vp, vp], None

    def line(self, pq, integer=False):
        dt = np.int32 if integer else np.float64
        a, out = np.ascontiguousarray(pq, dtype=dt).reshape(4), np.zeros(3, dtype=dt)
        (self.lib.ssdo_line_i if integer else self.lib.ssdo_line_d)(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def line_dets(self, l, o):
        a, b, out = np.ascontiguousarray(l, dtype=np.float64), np.ascontiguousarray(o, dtype=np.float64), np.zeros(3)
        self.lib.ssdo_line_dets_d(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def config(self, width, height):
        cfg = Config()
        self.lib.ssdo_default_config(C.byref(cfg), width, height)
        return cfg

    def calibration(self, world, cam):
        cal = Calibration()
        w = (C.c_double * 9)(*np.asarray(world, dtype=np.float64).reshape(9))
        c = (C.c_double * 9)(*np.asarray(cam, dtype=np.float64).reshape(9))
        rc = self.lib.ssdo_calibration_from_points(w, c, C.byref(cal))
        return rc, cal

    def risers(self, cfg, cal, xyz, tolerance=0.03, min_support=200):
        """extension (no reference counterpart): riser evidence -> list of Riser"""
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        out = (Riser * MAX_STEPS)()
        n = self.lib.ssdo_risers(C.byref(cfg), C.byref(cal), a.ctypes.data_as(C.c_void_p), tolerance, min_support, out)
        if n < 0:
            raise RuntimeError("ssdo_risers failed: %d" % n)
        return [out[i] for i in range(n)]

    def sort_perm(self, dist):
        d = np.ascontiguousarray(dist, dtype=np.float64)
        perm = np.zeros(len(d), dtype=np.int32)
        self.lib.ssdo_sort_perm(d.ctypes.data_as(C.c_void_p), len(d), perm.ctypes.data_as(C.c_void_p))
        return perm

    def deproject(self, intr, depth):
        a = np.ascontiguousarray(depth, dtype=np.uint16)
        out = np.empty(a.shape + (3,), dtype=np.float32)
        self.lib.ssdo_deproject(intr.fx, intr.fy, intr.ppx, intr.ppy, intr.depth_units, a.shape[1], a.shape[0],
                                a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def calibration_load(self, directory):
        w, c = np.zeros(9), np.zeros(9)
        rc = self.lib.ssdo_calibration_load(os.path.join(directory, "calibration-triangle").encode(),
                                            os.path.join(directory, "calibration-points").encode(),
                                            w.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        return rc, w.reshape(3, 3), c.reshape(3, 3)

    def process(self, cfg, cal, xyz, images=0, ground_images=False):
        """-> (Result, raw_images, closed_images, ground_raw, ground_closed)"""
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        n = cfg.width * cfg.height
        assert a.size == 3 * n
        res = Result()
        raw = np.zeros((images, cfg.height, cfg.width), dtype=np.uint8) if images else None
        closed = np.zeros((images, cfg.height, cfg.width), dtype=np.uint8) if images else None
        graw = np.zeros((cfg.height, cfg.width), dtype=np.uint8) if ground_images else None
        gclosed = np.zeros((cfg.height, cfg.width), dtype=np.uint8) if ground_images else None
        p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
        rc = self.lib.ssdo_process(C.byref(cfg), C.byref(cal), p(a), C.byref(res), p(raw), p(closed), images, p(graw), p(gclosed))
        if rc < 0:
            raise RuntimeError("ssdo_process failed: %d" % rc)
        return res, raw, closed, graw, gclosed

    def process_lean(self, cfg, cal, xyz):
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        steps = np.zeros((MAX_STEPS, 9), dtype=np.float64)
        status = C.c_int(0)
        n = self.lib.ssdo_process_lean(C.byref(cfg), C.byref(cal), a.ctypes.data_as(C.c_void_p),
                                       steps.ctypes.data_as(C.c_void_p), C.byref(status))
        return n, steps[:max(n, 0)], status.value

    def process_many(self, cfg, cal, frames, cpus, reps=1):
        """ssdo_process_many: the oracle on len(cpus) pinned host threads at once, thread t on a private copy of frames[t % len(frames)],
        `reps` frames each -> dict(wall_s, frames, frames_per_s, read_gb_per_s, steps)"""
        arrs = [np.ascontiguousarray(f, dtype=np.float32) for f in frames]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        cp = (C.c_int * len(cpus))(*[int(c) for c in cpus])
        wall, rd, steps = C.c_double(0.0), C.c_double(0.0), C.c_longlong(0)
        rc = self.lib.ssdo_process_many(C.byref(cfg), C.byref(cal), ptrs, len(arrs), cp, len(cpus), int(reps), C.byref(wall), C.byref(rd), C.byref(steps))
        if rc < 0:
            raise RuntimeError("ssdo_process_many failed: %d" % rc)
        n = len(cpus) * int(reps)
        return dict(wall_s=wall.value, frames=n, frames_per_s=n / wall.value if wall.value > 0 else 0.0, read_gb_per_s=rd.value, steps=steps.value)

    def close3x3(self, img):
        a = np.ascontiguousarray(img, dtype=np.uint8).copy()
        self.lib.ssdo_close3x3(a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0])
        return a

    def serialize(self, steps_ext):
        s = np.ascontiguousarray(steps_ext, dtype=np.float64).reshape(-1, 9)
        buf = C.create_string_buffer(LINE_CAP)
        self.lib.ssdo_serialize(len(s), s.ctypes.data_as(C.c_void_p), buf, LINE_CAP)
        return buf.value.decode()

    def quad_test(self, quad, pts):
        q = np.ascontiguousarray(quad, dtype=np.float64).reshape(8)
        p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
        out = np.zeros(len(p), dtype=np.uint8)
        rc = self.lib.ssdo_quad_test(q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), len(p), out.ctypes.data_as(C.c_void_p))
        return rc, out

    def hypot(self, a, b):
        return self.lib.ssdo_hypot(a, b)

    def best_line(self, pts):
        p = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
        out = np.zeros(3, dtype=np.int32)
        rc = self.lib.ssdo_best_line(p.ctypes.data_as(C.c_void_p), len(p), out.ctypes.data_as(C.c_void_p))
        return rc, out


class Ref:
    """The real reference's Stairs::serialize and QuadrilateralTest (oracle/_ref/libssd_ref.so)."""

    def __init__(self, lib):
        self.lib = lib
        lib.ssdref_serialize.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        lib.ssdref_quad_test.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.ssdref_load_triangle.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]
        lib.ssdref_configuration.argtypes = [C.c_void_p, C.c_void_p]
        lib.ssdref_configuration.restype = None
        for f in (lib.ssdref_line_d, lib.ssdref_line_i):
            f.argtypes, f.restype = [C.c_void_p, C.c_void_p], None
        lib.ssdref_line_dets_d.argtypes, lib.ssdref_line_dets_d.restype = [C.c_void_p] * 3, None

    def line(self, pq, integer=False):
        """LineCoordinates<T>(p, q) of the reference (types.h:140-158) -> its three coefficients"""
        dt = np.int32 if integer else np.float64
        a, out = np.ascontiguousarray(pq, dtype=dt).reshape(4), np.zeros(3, dtype=dt)
        (self.lib.ssdref_line_i if integer else self.lib.ssdref_line_d)(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def line_dets(self, l, o):
        """LineCoordinates<double>::det / detx / dety (types.h:128-139)"""
        a, b, out = np.ascontiguousarray(l, dtype=np.float64), np.ascontiguousarray(o, dtype=np.float64), np.zeros(3)
        self.lib.ssdref_line_dets_d(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def configuration(self):
        """the reference's default-constructed Configuration (configuration.h:27-52) -> (9 doubles, (width, height))"""
        out = np.zeros(9)
        wh = np.zeros(2, dtype=np.int32)
        self.lib.ssdref_configuration(out.ctypes.data_as(C.c_void_p), wh.ctypes.data_as(C.c_void_p))
        return out, (int(wh[0]), int(wh[1]))

    def serialize(self, steps_ext):
        s = np.ascontiguousarray(steps_ext, dtype=np.float64).reshape(-1, 9)
        buf = C.create_string_buffer(LINE_CAP)
        self.lib.ssdref_serialize(len(s), s.ctypes.data_as(C.c_void_p), buf, LINE_CAP)
        return buf.value.decode()

    def load_triangle(self, directory):
        w = np.zeros(9)
        side = C.c_int(0)
        rc = self.lib.ssdref_load_triangle(directory.encode(), w.ctypes.data_as(C.c_void_p), C.byref(side))
        return rc, w.reshape(3, 3), side.value

    def resave_triangle(self, dir_in, dir_out):
        """CalibrationTriangle::load() in dir_in, ::save() in dir_out (the reference's own writer)"""
        self.lib.ssdref_resave_triangle.argtypes = [C.c_char_p, C.c_char_p]
        return self.lib.ssdref_resave_triangle(dir_in.encode(), dir_out.encode())

    def quad_test(self, quad, pts):
        q = np.ascontiguousarray(quad, dtype=np.float64).reshape(8)
        p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
        out = np.zeros(len(p), dtype=np.uint8)
        rc = self.lib.ssdref_quad_test(q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), len(p), out.ctypes.data_as(C.c_void_p))
        return rc, out


def load_oracle():
    if not os.path.exists(ORACLE_LIB):
        build_oracle()
    return Oracle(C.CDLL(ORACLE_LIB))


def load_ref():
    if not os.path.exists(REF_LIB):
        return None
    return Ref(C.CDLL(REF_LIB))


def to_oracle_config(cfg):
    """product Config -> oracle Config (same field values)."""
    o = Config()
    for f, _ in Config._fields_:
        setattr(o, f, getattr(cfg, f))
    return o


def to_oracle_calibration(cal):
    o = Calibration()
    C.memmove(C.byref(o), C.byref(cal), C.sizeof(Calibration))
    return o
