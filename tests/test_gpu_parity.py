"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import parity
import scenes

pytestmark = pytest.mark.gpu

SCENES = sorted(scenes.scene_params().keys())


def _setup(ssd, name, frames=1):
    sc = scenes.make(ssd, name)
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(sc.width, sc.height, max_frames_per_batch=frames)
    return sc, trans, cfg


@pytest.mark.parametrize("name", SCENES)
def test_frame_parity_all_intermediates(ssd, oracle, gpu_device, name):
    """Every intermediate (histogram, peaks, plateau table, raw + closed images, scans, lines, probe points,
    corners, in-quad counts, mean z) and the serialized line, frame by frame."""
    sc, trans, cfg = _setup(ssd, name)
    xyz = ssd.synth_host([sc])[0]
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
    det.close()
    assert rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT
    assert rep.get("max_corner_err", 0.0) == 0.0


@pytest.mark.parametrize("name", SCENES)
def test_frame_results_without_debug_capture(ssd, oracle, gpu_device, name):
    """The production path (no debug capture: closing and scans only inside the bounding boxes of the raw bits, images
    never copied out) against the oracle's result: status, steps, heights, corners, serialized line."""
    sc, trans, cfg = _setup(ssd, name)
    xyz = ssd.synth_host([sc])[0]
    det = ssd.Detector(cfg, trans, gpu_device)
    fr = det.process_host(xyz)[0]
    det.close()
    rep = parity.check_results_only(ssd, oracle, cfg, trans.constants, xyz, fr)
    assert rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT
    assert rep.get("max_corner_err", 0.0) == 0.0


@pytest.mark.parametrize("name", ["vga_3steps_noise2mm", "xga_config1", "xga_yaw_p8", "fhd_3steps_clean", "ragged_427x321_yaw", "ragged_600x450", "vga_yaw_outliers", "xga_low_camera"])
def test_ground_raster_is_the_strips_the_bottom_scan_reads(ssd, oracle, gpu_device, name):
    """Outside image capture k_inquad sets, of the ground points inside the ground quadrilateral (pointcloud.cpp:530-531), only the
    pixels detectFrontEdge can see: the columns within two of a scan column W/2 + 50 k (BottomScanner, segmentation.cpp:159-241;
    the closing reaches two pixels) and the rows from H/2 - 1 down — and of those (round 4) only what the scan's bottom-most lit
    pixel can depend on: per strip, the rows from two above the bottom-most pixel of the strip's CENTRE column down (which pixels
    above that line are written is a matter of the blocks' timing).  The workspace image after ssd_enqueue_stages(.. INQUAD) must
    lie inside the oracle's raw ground image restricted to the strips and contain all of it from that line down; with image capture
    on it must be the whole image; and after a full run it must be zero again (k_final clears what k_inquad set)."""
    sc, trans, cfg = _setup(ssd, name)
    W, H = sc.width, sc.height
    xyz = ssd.synth_host([sc])[0]
    res, _, _, graw, _ = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz, images=0, ground_images=True)
    assert res.first_valid_ind >= 0 and res.ground_ind >= 0 and graw.any()
    xs = np.arange(W)
    x0 = (W // 2) % 50
    in_strip = ((xs + 2 - x0) % 50) <= 4
    mask = np.zeros((H, W), dtype=bool)
    mask[H // 2 - 1:, :] = in_strip[None, :]
    buf = ssd.DeviceBuffer(W * H * 12, gpu_device)
    buf.upload(xyz)
    det = ssd.Detector(cfg, trans, gpu_device)
    upto_inquad = ssd.STAGE_ALL & ~64
    det.enqueue(buf.ptr, 1, stages=upto_inquad)
    got = det.ground_image_raw(0)
    want = np.where(mask, graw, 0).astype(np.uint8)
    assert not (got & ~want).any(), "%d pixels set that are not ground pixels of a strip" % int(((got != 0) & (want == 0)).sum())
    must = np.zeros((H, W), dtype=bool)                     # what the bottom scan can depend on
    for xj in range(x0 - 50, W + 50, 50):
        cols = [c for c in range(xj - 2, xj + 3) if 0 <= c < W]
        if not cols:
            continue
        centre = np.nonzero(want[:, xj])[0] if 0 <= xj < W else np.array([], dtype=int)
        first = max(int(centre.max()) - 2, 0) if centre.size else 0          # no centre pixel: nothing may be left out
        must[first:, cols] = True
    missing = (want != 0) & must & (got == 0)
    assert not missing.any(), "%d pixels the bottom scan can depend on are missing" % int(missing.sum())
    assert 0 < int((got != 0).sum()) <= int((want != 0).sum()) < int((graw != 0).sum()) // 4
    det.enqueue(buf.ptr, 1)                                   # the partial run left bits behind: the library clears them first
    fr = det.fetch_list(1)[0]
    assert not det.ground_image_raw(0).any()
    parity.check_results_only(ssd, oracle, cfg, trans.constants, xyz, fr)
    det.set_debug(True, images=True)
    det.enqueue(buf.ptr, 1, stages=upto_inquad)
    assert np.array_equal(det.ground_image_raw(0), graw)
    det.set_debug(False)
    det.enqueue(buf.ptr, 1)
    assert bytes(det.fetch_list(1)[0]) == bytes(fr) and not det.ground_image_raw(0).any()
    det.close()
    buf.free()


def test_scene_set_covers_the_interesting_outcomes(ssd, oracle):
    """Guards the fixture set itself (oracle only): it must contain N=0 lines, ground+steps, invalid plateaus."""
    outcomes = set()
    for name in ("xga_config1", "xga_no_stairs", "vga_empty", "xga_wide", "vga_8steps_outliers"):
        sc, trans, cfg = _setup(ssd, name)
        xyz = ssd.synth_host([sc])[0]
        n, steps, status = oracle.process_lean(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz)
        outcomes.add((name, n))
    d = dict(outcomes)
    assert d["xga_config1"] == 4
    assert d["xga_no_stairs"] == 0 and d["vga_empty"] == 0
    for name in ("vga_yaw40_wide_throws", "vga_yaw50_throws"):
        sc, trans, cfg = _setup(ssd, name)
        n, steps, status = oracle.process_lean(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), ssd.synth_host([sc])[0])
        assert status & ob.ST_THROW and n == 0


def test_device_generator_matches_host_generator(ssd, gpu_device):
    sc_list = [scenes.make(ssd, "vga_yaw_outliers"), scenes.make(ssd, "vga_3steps_noise2mm"), scenes.make(ssd, "vga_8steps_outliers")]
    host = ssd.synth_host(sc_list)
    buf = ssd.DeviceBuffer(host.nbytes, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    dev = buf.download(host.nbytes, dtype=np.float32).reshape(host.shape)
    buf.free()
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))


def test_batch_equals_single_frames_and_workspace_is_clean(ssd, oracle, gpu_device):
    """A batch gives, frame by frame, what single-frame calls give; running it twice (images must have been
    cleared by their consumers) gives the same again; results equal the oracle's."""
    names = ["vga_3steps_clean", "vga_3steps_noise2mm", "vga_8steps_outliers", "vga_empty", "vga_yaw_outliers", "vga_3steps_clean"]
    sc_list = [scenes.make(ssd, n) for n in names]
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(640, 480, max_frames_per_batch=4)      # 6 frames -> two chunks (4 + 2)
    xyz = ssd.synth_host(sc_list)
    det = ssd.Detector(cfg, trans, gpu_device)
    r1 = det.process_host(xyz)
    r2 = det.process_host(xyz)
    singles = [det.process_host(xyz[i])[0] for i in range(len(names))]
    for i in range(len(names)):
        assert bytes(r1[i]) == bytes(r2[i]), "second batch differs: workspace not clean (frame %d)" % i
        assert bytes(r1[i]) == bytes(singles[i]), "batch differs from single-frame call (frame %d)" % i
        parity.check_results_only(ssd, oracle, cfg, trans.constants, xyz[i], r1[i])
    det.close()


def test_device_resident_enqueue_fetch(ssd, oracle, gpu_device):
    """The asynchronous entry points on frames generated in HBM, with a padded frame stride."""
    sc_list = scenes.batch_scenes(ssd, 640, 480, 5)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(640, 480, max_frames_per_batch=8)
    frame_bytes = 640 * 480 * 12
    stride = frame_bytes + 256
    buf = ssd.DeviceBuffer(stride * 5, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, stride_bytes=stride, device=gpu_device)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.set_timing(True)
    det.enqueue(buf.ptr, 5, stride_bytes=stride)
    res = det.fetch_list(5)
    times = det.stage_times_ms()
    assert all(t >= 0.0 for t in times.values()) and times["hist"] > 0.0
    host = ssd.synth_host(sc_list)
    for i in range(5):
        parity.check_results_only(ssd, oracle, cfg, trans.constants, host[i], res[i])
    det.close()
    buf.free()


def test_unaligned_frames_take_the_12_byte_load_path(ssd, oracle, gpu_device):
    """A frame stride that is not a multiple of 16 bytes makes the kernels fall back from 16-byte to
    12-byte loads; results must not change."""
    sc_list = scenes.batch_scenes(ssd, 640, 480, 3, base_seed=77)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(640, 480, max_frames_per_batch=4)
    frame_bytes = 640 * 480 * 12
    host = ssd.synth_host(sc_list)
    det = ssd.Detector(cfg, trans, gpu_device)
    want = det.process_host(host)
    stride = frame_bytes + 4
    buf = ssd.DeviceBuffer(stride * 3 + 16, gpu_device)
    for i in range(3):
        buf.upload(host[i], offset=4 + i * stride)          # base pointer 4 (not 16) byte aligned as well
    det.enqueue(buf.ptr + 4, 3, stride_bytes=stride)
    got = det.fetch_list(3)
    assert [bytes(x) for x in got] == [bytes(x) for x in want]
    det.close()
    buf.free()


def test_device_hypot_matches_libm(ssd, oracle, gpu_device):
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.integers(-2000, 2000, 4000).astype(np.float64), rng.standard_normal(4000), rng.standard_normal(2000) * 1e-3])
    b = np.concatenate([rng.integers(-2000, 2000, 4000).astype(np.float64), rng.standard_normal(4000), rng.standard_normal(2000) * 1e3])
    out = np.zeros_like(a)
    rc = ssd.hooks_lib().ssd_test_hypot_device(gpu_device, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), len(a))
    assert rc == 0
    want = np.array([oracle.hypot(float(x), float(y)) for x, y in zip(a, b)])
    assert np.array_equal(out, want)


def test_full_size_properties_xga_batch(ssd, oracle, gpu_device):
    """BASELINE.json config 3 shape at reduced count: size-independent properties on XGA frames in HBM —
    determinism (bitwise), permutation equivariance over frames, histogram mass = in-range count, and an
    oracle spot check of every 8th frame."""
    n = 24
    sc_list = scenes.batch_scenes(ssd, 1024, 768, n, base_seed=5000)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(1024, 768, max_frames_per_batch=n)
    frame_bytes = 1024 * 768 * 12
    buf = ssd.DeviceBuffer(frame_bytes * n, gpu_device)
    ssd.synth_device(sc_list, buf.ptr, device=gpu_device)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.enqueue(buf.ptr, n)
    r1 = det.fetch_list(n)
    det.enqueue(buf.ptr, n)
    r2 = det.fetch_list(n)
    assert [bytes(x) for x in r1] == [bytes(x) for x in r2]
    perm = list(reversed(range(n)))
    ssd.synth_device([sc_list[i] for i in perm], buf.ptr, device=gpu_device)
    det.enqueue(buf.ptr, n)
    r3 = det.fetch_list(n)
    assert [bytes(r3[j]) for j in range(n)] == [bytes(r1[perm[j]]) for j in range(n)]
    assert sum(1 for r in r1 if r.n_steps >= 3) >= n // 2
    host = ssd.synth_host([sc_list[i] for i in range(0, n, 8)])
    for k, i in enumerate(range(0, n, 8)):
        parity.check_results_only(ssd, oracle, cfg, trans.constants, host[k], r1[i])
    det.close()
    buf.free()


def test_detect_stairs_driver_prints_the_reference_line(ssd, oracle, gpu_device):
    """The C++ surface (stairs::Pointcloud::process via lib/detect-stairs-amd, the detect-stairs.cpp loop) prints
    exactly the line the oracle derives for the same synthetic frame."""
    import math
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(ssd.LIB_PATH), "detect-stairs-amd")
    assert os.path.exists(exe), "driver not built"
    out = subprocess.run([exe, "--width", "640", "--height", "480", "--frames", "2", "--steps", "3", "--seed", "77"],
                         check=True, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
    assert len(out) == 2
    for f in range(2):
        sc = ssd.make_scene(640, 480, n_steps=3, seed=77 + f)      # the driver's makeScene() defaults
        trans = ssd.GeometricTransformation(*ssd.calibration_points(sc, world_offset=(0.0, 0.0, 0.0)))
        cfg = ssd.default_config(640, 480)
        res, *_ = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), ssd.synth_host([sc])[0])
        assert out[f] == res.line.decode()


DEPTH_SCENES = ["vga_3steps_noise2mm", "xga_config1", "xga_yaw_p8", "vga_8steps_outliers", "fhd_3steps_noise2mm", "xga_no_stairs", "ragged_600x450"]


@pytest.mark.parametrize("name", DEPTH_SCENES)
def test_depth_input_parity_all_intermediates(ssd, oracle, gpu_device, name):
    """16-bit depth frames deprojected on the fly by the kernels vs the oracle deprojecting first: every intermediate."""
    sc, trans, cfg = _setup(ssd, name)
    depth = ssd.synth_depth_host([sc])[0]
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, depth, images=True, depth_intr=ssd.intrinsics_for_scene(sc))
    det.close()
    assert rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT


def test_new_intrinsics_wait_for_the_depth_batches_in_flight(ssd, gpu_device):
    """ssd_set_intrinsics rewrites the deprojection maps the kernels of depth batches read; with several
    workspaces those batches run on streams of the handle's own, which a copy on the default stream does not wait for.  Three
    batches enqueued, other intrinsics set at once, then the fetches: every batch must be what the FIRST intrinsics give (the
    call waits for the lanes), and a batch after the call what the SECOND give."""
    import ctypes as C
    n, W, H = 48, 640, 480
    sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=654, rng_seed=2)
    trans = ssd.transformation_for_scene(sc_list[0])
    intr = ssd.intrinsics_for_scene(sc_list[0])
    other = ssd.Intrinsics(intr.fx * 1.25, intr.fy * 0.8, intr.ppx + 7.0, intr.ppy - 5.0, intr.depth_units)
    buf = ssd.DeviceBuffer(n * W * H * 2, gpu_device)
    ssd.synth_depth_device(sc_list, buf.ptr, device=gpu_device)
    one = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n), trans, gpu_device)
    want = {}
    for key, it in (("first", intr), ("second", other)):
        one.set_intrinsics(it)
        one.enqueue_depth(buf.ptr, n)
        want[key] = [bytes(x) for x in one.fetch_list(n)]
    one.close()
    assert want["first"] != want["second"]
    det = ssd.Detector(ssd.default_config(W, H, max_frames_per_batch=n, batches_in_flight=3), trans, gpu_device)
    for rep in range(3):
        det.set_intrinsics(intr)
        for _ in range(3):
            det.enqueue_depth(buf.ptr, n)
        det.set_intrinsics(other)                      # while the three batches are still running
        got = [[bytes(x) for x in det.fetch(n, back=b)] for b in (2, 1, 0)]
        assert all(g == want["first"] for g in got), "a batch in flight saw the new maps (round %d)" % rep
        det.enqueue_depth(buf.ptr, n)
        assert [bytes(x) for x in det.fetch_list(n)] == want["second"]
    det.close()
    buf.free()


def test_depth_path_equals_float_path_on_the_deprojected_cloud(ssd, gpu_device):
    """Same handle, same frames: depth input (device-generated, in HBM) and float input (host-deprojected) give bitwise equal results."""
    sc_list = scenes.batch_scenes(ssd, 640, 480, 6, base_seed=321)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(640, 480, max_frames_per_batch=8)
    intr = ssd.intrinsics_for_scene(sc_list[0])
    depth = ssd.synth_depth_host(sc_list)
    buf = ssd.DeviceBuffer(depth.nbytes, gpu_device)
    ssd.synth_depth_device(sc_list, buf.ptr, device=gpu_device)
    assert np.array_equal(buf.download(depth.nbytes, dtype=np.uint16).reshape(depth.shape), depth)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.set_intrinsics(intr)
    det.enqueue_depth(buf.ptr, 6)
    got = det.fetch_list(6)
    want = det.process_host(np.stack([ssd.deproject_host(intr, d) for d in depth]))
    assert [bytes(x) for x in got] == [bytes(x) for x in want]
    assert sum(1 for r in got if r.n_steps >= 3) >= 3
    det.close()
    buf.free()


@pytest.mark.parametrize("variant", ["coarse_bins", "shifted_range", "fine_bins"])
def test_non_default_configuration(ssd, oracle, gpu_device, variant):
    """The reference's Configuration is compile-time (configuration.h:27-52); here it is a run-time struct: other
    measuring ranges, bin widths and thresholds must give the oracle's results too (derived constants of
    pointcloud.cpp:99-106 included)."""
    sc = scenes.make(ssd, "xga_yaw_p8")
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(sc.width, sc.height, max_frames_per_batch=1)
    if variant == "coarse_bins":
        cfg.height_interval = 0.02                      # 61 bins
        cfg.min_height_above_ground = 0.08
    elif variant == "shifted_range":
        cfg.x_min, cfg.x_max = -0.45, 0.7
        cfg.y_min, cfg.y_max = 0.25, 1.15
        cfg.z_min, cfg.z_max = -0.05, 0.9
        cfg.min_step_depth = 0.15
    else:
        cfg.height_interval = 0.0095                    # 127 bins: just inside SSD_MAX_BINS
    xyz = ssd.synth_host([sc])[0]
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
    det.close()
    assert rep["n_steps"] >= 2


def test_hip_path_reproduces_the_surveys_observed_run(ssd, gpu_device, tmp_path):
    """SURVEY.md Appendix A's frame through the HIP path, against the values the survey observed from the reference
    (a sanity anchor, see tests/survey_anchor.py): corners bit-exact, heights to 1e-9 m, the line byte for byte."""
    import survey_anchor as sa
    xyz, cam = sa.frame(tmp_path)
    trans = ssd.GeometricTransformation(sa.WORLD_POINTS.reshape(3, 3), cam.reshape(3, 3))
    det = ssd.Detector(ssd.default_config(1024, 768, max_frames_per_batch=1), trans, gpu_device)
    fr = det.process_host(xyz)[0]
    det.close()
    assert (fr.n_steps, fr.status) == (4, 0)
    for i in range(4):
        assert abs(fr.steps[i].height - sa.OBSERVED[i, 0]) <= parity.TOL_HEIGHT
        assert list(fr.steps[i].quad) == list(sa.OBSERVED[i, 1:])
    assert ssd.Stairs(fr).serialize() == sa.OBSERVED_LINE


@pytest.mark.gpu
def test_cells_classified_as_inside_hold_only_points_the_quadrilateral_test_accepts(ssd, gpu_device):
    """k_inquad skips a tread cell whose box on K1's 256 x 256 grid lies inside all four edges of the quadrilateral
    (build_grid_segs / grid_box_inside).  Property: every point of such a box — corners, edge midpoints, random interior
    points, nudged 1e-9 m outwards as K1's truncation may — passes the kernels' QuadrilateralTest (itself pinned to the
    reference's goldens above); and the shortcut is not vacuous: it accepts most boxes well inside an ordinary tread.
    Quadrilaterals: the reference goldens' (incl. the degenerate ones that throw) + random turned / sheared treads."""
    import json
    import os
    rng = np.random.default_rng(20260)
    x_min, y_min, span = -0.6, 0.1, 1.2
    box = 256.0 / span
    quads = []
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_quadtest.json")))
    for c in cases:
        quads.append(np.array([float.fromhex(x) for x in c["quad"]]).reshape(4, 2))
    for _ in range(60):
        yaw = np.radians(rng.uniform(-45, 45))
        w, d = rng.uniform(0.3, 1.0), rng.uniform(0.1, 0.4)
        cx, cy = rng.uniform(-0.2, 0.2), rng.uniform(0.4, 1.0)
        base = np.array([[-w / 2, -d / 2], [w / 2, -d / 2], [-w / 2, d / 2], [w / 2, d / 2]])     # fl, fr, bl, br
        base += rng.normal(0, 0.02, base.shape)                                                    # sheared / trapezoid
        rot = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        quads.append(base @ rot.T + [cx, cy])
    accepted = usable_quads = 0
    for quad in quads:
        n = 600
        x0 = rng.integers(0, 250, n); y0 = rng.integers(0, 250, n)
        x1 = np.minimum(255, x0 + rng.integers(0, 24, n)); y1 = np.minimum(255, y0 + rng.integers(0, 6, n))
        boxes = np.stack([x0, x1, y0, y1], 1)
        usable, inside = ssd.grid_boxes_device(quad, x_min, y_min, box, box, boxes, gpu_device)
        if not usable:
            assert not inside.any()
            continue
        usable_quads += 1
        sel = boxes[inside != 0]
        accepted += len(sel)
        if len(sel) == 0:
            continue
        # the world rectangle a box stands for, widened as K1's truncation may have misplaced a point
        e = 1.0e-9
        X0, X1 = x_min + sel[:, 0] / box - e, x_min + (sel[:, 1] + 1) / box + e
        Y0, Y1 = y_min + sel[:, 2] / box - e, y_min + (sel[:, 3] + 1) / box + e
        u = rng.uniform(0, 1, (len(sel), 12, 2))
        u[:, 0] = [0, 0]; u[:, 1] = [1, 0]; u[:, 2] = [0, 1]; u[:, 3] = [1, 1]
        u[:, 4] = [0.5, 0]; u[:, 5] = [0.5, 1]; u[:, 6] = [0, 0.5]; u[:, 7] = [1, 0.5]
        pts = np.stack([X0[:, None] + u[:, :, 0] * (X1 - X0)[:, None], Y0[:, None] + u[:, :, 1] * (Y1 - Y0)[:, None]], 2).reshape(-1, 2)
        err, ok = ssd.quad_test_device(quad, pts, gpu_device)
        assert err == 0
        assert ok.all(), ("a point of a box classified as inside fails the quadrilateral test", quad.tolist(), pts[ok == 0][:3].tolist())
    assert usable_quads >= 50 and accepted > 2000, (usable_quads, accepted)


@pytest.mark.parametrize("name", SCENES)
def test_riser_evidence_matches_the_cpu_statement(ssd, oracle, gpu_device, name):
    """Extension (no reference counterpart): the evidence of the vertical faces gathered by
    k_risers against oracle/'s statement of the same spec; and switching it on must not change the reference results."""
    sc, trans, cfg = _setup(ssd, name)
    xyz = ssd.synth_host([sc])[0]
    det = ssd.Detector(cfg, trans, gpu_device)
    plain = bytes(det.process_host(xyz)[0])
    det.set_risers(True, tolerance=0.03, min_support=200)
    fr = det.process_host(xyz)[0]
    assert bytes(fr) == plain
    dev = det.fetch_risers(1)[0]
    det.close()
    ora = oracle.risers(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz, 0.03, 200)
    rep = parity.compare_risers(dev, ora)
    if name in ("xga_config1", "xga_yaw_p8", "fhd_3steps_clean"):
        assert rep["risers_detected"] == 3


def test_risers_of_a_host_fed_batch_cover_every_slice(ssd, oracle, gpu_device):
    """ssd_process_host cuts a batch into slices of 32 frames; the riser buffer on the device holds one enqueue's.  The
    library collects the risers slice by slice (before, ssd_fetch_risers silently returned the last slice's):
    all 70 frames' risers must equal the oracle's, and asking for more frames than the call processed is an error."""
    n, W, H = 70, 640, 480
    sc_list = scenes.batch_scenes(ssd, W, H, n, base_seed=33000, rng_seed=6)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(W, H, max_frames_per_batch=64)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.set_risers(True, tolerance=0.02, min_support=300)
    xyz = ssd.synth_host(sc_list)
    for rep in range(2):
        res = det.process_host(xyz)
        rs = det.fetch_risers(n)
        assert len(res) == n and len(rs) == n
        ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
        for i in list(range(0, n, 9)) + [31, 32, 63, 64, 69]:
            parity.compare_risers(rs[i], oracle.risers(ocfg, ocal, xyz[i], 0.02, 300))
    with pytest.raises(ssd.SsdError, match="exceeds"):
        det.fetch_risers(n + 1)
    # a device-resident call afterwards: its own risers again
    buf = ssd.DeviceBuffer(W * H * 12 * 3, gpu_device)
    buf.upload(xyz[5:8])
    det.enqueue(buf.ptr, 3)
    for i, fr in enumerate(det.fetch_risers(3)):
        parity.compare_risers(fr, oracle.risers(ocfg, ocal, xyz[5 + i], 0.02, 300))
    with pytest.raises(ssd.SsdError, match="exceeds"):
        det.fetch_risers(4)
    buf.free()
    det.close()


def test_risers_in_a_batch_and_with_depth_input(ssd, oracle, gpu_device):
    sc_list = scenes.batch_scenes(ssd, 640, 480, 6, base_seed=555)
    trans = ssd.transformation_for_scene(sc_list[0])
    cfg = ssd.default_config(640, 480, max_frames_per_batch=6)
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.set_risers(True, tolerance=0.02, min_support=500)
    host = ssd.synth_host(sc_list)
    det.process_host(host)
    for i, fr in enumerate(det.fetch_risers(6)):
        parity.compare_risers(fr, oracle.risers(ocfg, ocal, host[i], 0.02, 500))
    intr = ssd.intrinsics_for_scene(sc_list[0])
    det.set_intrinsics(intr)
    depth = ssd.synth_depth_host(sc_list)
    det.process_depth_host(depth)
    for i, fr in enumerate(det.fetch_risers(6)):
        parity.compare_risers(fr, oracle.risers(ocfg, ocal, oracle.deproject(intr, depth[i]), 0.02, 500))
    det.close()


def test_results_travel_with_their_enqueue(ssd, oracle, gpu_device):
    """Two batches in flight: the second is enqueued before the first one's results are read (ssd_fetch_back)."""
    a = scenes.batch_scenes(ssd, 640, 480, 3, base_seed=31)
    b = scenes.batch_scenes(ssd, 640, 480, 2, base_seed=77)
    trans = ssd.transformation_for_scene(a[0])
    det = ssd.Detector(ssd.default_config(640, 480, max_frames_per_batch=3), trans, gpu_device)
    want_a = [bytes(r) for r in det.process_host(ssd.synth_host(a))]
    want_b = [bytes(r) for r in det.process_host(ssd.synth_host(b))]
    assert want_a[:2] != want_b
    fb = 640 * 480 * 12
    buf_a, buf_b = ssd.DeviceBuffer(3 * fb, gpu_device), ssd.DeviceBuffer(2 * fb, gpu_device)
    ssd.synth_device(a, buf_a.ptr, device=gpu_device)
    ssd.synth_device(b, buf_b.ptr, device=gpu_device)
    for _ in range(3):                                     # both slot parities
        det.enqueue(buf_a.ptr, 3)
        det.enqueue(buf_b.ptr, 2)
        assert [bytes(r) for r in det.fetch(3, back=1)] == want_a
        assert [bytes(r) for r in det.fetch(2, back=0)] == want_b
        det.enqueue(buf_a.ptr, 3)
        assert [bytes(r) for r in det.fetch(2, back=1)] == want_b
        assert [bytes(r) for r in det.fetch(3)] == want_a
    with pytest.raises(ssd.SsdError):
        det.fetch(3, back=2)
    with pytest.raises(ssd.SsdError):
        det.fetch(3, back=1)                               # that enqueue held 2 frames
    det.close(); buf_a.free(); buf_b.free()


def test_host_batches_larger_than_the_workspace_are_processed_in_chunks(ssd, oracle, gpu_device):
    """ssd_process_host with more frames than max_frames_per_batch loops over chunks (also the 16-bit depth entry)."""
    sc_list = scenes.batch_scenes(ssd, 640, 480, 5, base_seed=909)
    trans = ssd.transformation_for_scene(sc_list[0])
    host = ssd.synth_host(sc_list)
    big = ssd.Detector(ssd.default_config(640, 480, max_frames_per_batch=8), trans, gpu_device)
    want = [bytes(r) for r in big.process_host(host)]
    small = ssd.Detector(ssd.default_config(640, 480, max_frames_per_batch=2), trans, gpu_device)
    assert [bytes(r) for r in small.process_host(host)] == want
    intr = ssd.intrinsics_for_scene(sc_list[0])
    depth = ssd.synth_depth_host(sc_list)
    big.set_intrinsics(intr)
    small.set_intrinsics(intr)
    assert [bytes(r) for r in small.process_depth_host(depth)] == [bytes(r) for r in big.process_depth_host(depth)]
    big.close()
    small.close()


def test_hip_path_reproduces_39_further_lines_of_the_surveys_probe(ssd, gpu_device, tmp_path):
    """The same 39 anchor lines (tests/golden/survey_probe_lines.json) through the HIP path, production mode."""
    import survey_anchor as sa
    dets = {}
    for c in sa.probe_cases():
        xyz, cam = sa.frame(tmp_path, case=c)
        trans = ssd.GeometricTransformation(sa.WORLD_POINTS.reshape(3, 3), cam.reshape(3, 3))
        det = ssd.Detector(ssd.default_config(c["width"], c["height"], max_frames_per_batch=1), trans, gpu_device)
        fr = det.process_host(xyz)[0]
        det.close()
        assert fr.status == 0
        assert ssd.Stairs(fr).serialize() == c["line"], c


def test_device_sort_restatement_equals_std_sort(ssd, oracle, gpu_device):
    rng = np.random.default_rng(5)
    for trial in range(60):
        n = int(rng.integers(1, 300))
        d = rng.integers(0, 6, n).astype(float) if trial % 2 else np.round(rng.uniform(0, 3, n), 1)
        assert np.array_equal(oracle.sort_perm(d), ssd.sort_perm(d, device=gpu_device))


def test_device_quadrilateral_test_against_the_reference_goldens(ssd, gpu_device):
    """PINNED: the kernels' QuadrilateralTest (cell map, constant cell, throw codes) against vectors produced by the
    reference's own quadrilateralTest.cpp (tests/golden/ref_quadtest.json, generator make_ref_goldens.py)."""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_quadtest.json")))
    seen = set()
    for c in cases:
        quad = np.array([float.fromhex(x) for x in c["quad"]]).reshape(4, 2)
        pts = np.array([float.fromhex(x) for x in c["pts"]]).reshape(-1, 2)
        err, inside = ssd.quad_test_device(quad, pts, gpu_device)
        assert err == c["rc"], (err, c["rc"])
        seen.add(err)
        if err == 0:
            assert "".join(str(int(v)) for v in inside) == c["inside"]
    assert 0 in seen and -1 in seen


def test_randomised_sweep_small(ssd, gpu_device):
    """tools/fuzz.py at test size (round 4: 28 random poses x 96 frames, 12 at FHD = 2100 frames; round 3: 8 x 32): four
    resolutions, depth input, non-default configurations, riser evidence, handles with one and with three workspaces — every
    frame of the batch path against the oracle."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py"), "28", "96", "4242", "mixed"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    summary = json.loads(p.stdout.strip().splitlines()[-1])
    assert summary["mismatches"] == 0 and summary["frames"] >= 2000


def test_one_handle_through_changing_calls_leaves_nothing_behind(ssd, oracle, gpu_device):
    """The per-frame device state is not zeroed in front of a call (k_peaks clears K1's accumulators as it takes them, every
    other field is written before it is read): one handle is driven through calls that differ in everything the state holds —
    8 steps with outliers, then empty frames, no stairs, throwing quadrilaterals, fewer frames than before, more than before,
    a K1-only call in between (which must trigger the zeroing), risers switched on and off — and every result must be, byte for
    byte, what a fresh handle gives for the same frames, and the oracle's."""
    W, H = 640, 480
    groups = [["vga_8steps_outliers", "vga_yaw_outliers", "vga_3steps_clean", "vga_8steps_outliers", "vga_3steps_noise2mm", "vga_yaw30_narrow"],
              ["vga_empty", "vga_empty"],
              ["vga_yaw40_wide_throws", "vga_yaw50_throws", "vga_3steps_clean"],
              ["vga_3steps_noise2mm"],
              ["vga_yaw_outliers", "vga_8steps_outliers", "vga_empty", "vga_yaw50_throws", "vga_3steps_clean", "vga_yaw30_narrow", "vga_8steps_outliers", "vga_3steps_noise2mm"],
              ["fuzz_border_closing_0", "vga_3steps_clean"]]
    first = scenes.make(ssd, groups[0][0])
    trans = ssd.transformation_for_scene(first)
    cfg = ssd.default_config(W, H, max_frames_per_batch=8)
    det = ssd.Detector(cfg, trans, gpu_device)
    buf = ssd.DeviceBuffer(8 * W * H * 12, gpu_device)
    for round_ in range(2):
        for gi, names in enumerate(groups):
            sc_list = [scenes.make(ssd, n) for n in names]
            xyz = ssd.synth_host(sc_list)
            n = len(names)
            risers = (gi + round_) % 3 == 1
            det.set_risers(risers, 0.03, 200)
            if gi == 3:
                # K1 alone on more frames than the next call takes: its accumulators stay dirty, the next call has to zero them
                buf.upload(ssd.synth_host([scenes.make(ssd, "vga_8steps_outliers")] * 5))
                det.enqueue(buf.ptr, 5, stages=ssd.STAGE_HIST)
            buf.upload(xyz)
            det.enqueue(buf.ptr, n)
            got = det.fetch_list(n)
            fresh_det = ssd.Detector(cfg, trans, gpu_device)
            fresh_det.set_risers(risers, 0.03, 200)
            fresh = fresh_det.process_host(xyz)
            for i in range(n):
                assert bytes(got[i]) == bytes(fresh[i]), "round %d, call %d, frame %d (%s): differs from a fresh handle" % (round_, gi, i, names[i])
                parity.check_results_only(ssd, oracle, cfg, trans.constants, xyz[i], got[i])
            if risers:
                a, b = det.fetch_risers(n), fresh_det.fetch_risers(n)
                for i in range(n):
                    assert bytes(a[i]) == bytes(b[i]), "round %d, call %d, frame %d: riser evidence differs from a fresh handle" % (round_, gi, i)
            fresh_det.close()
    det.close()
    buf.free()
