"""GPU parity on hand-built clouds that force the reference's quirks and the histogram / peak edge cases
(thresholds of pointcloud.cpp:243-256, non-finite input), with a calibration that is
the identity rotation plus a shift of -0.5 m in z (camera z = world z + 0.5 > 0, so that negative world heights exist);
`cloud()` takes world coordinates."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

W, H = 640, 480


def cloud(planes, extra=None, seed=0):
    """planes: list of (z, n_points, (x0, x1), (y0, y1)) -> float32 [H, W, 3]; points of a plane lie on a regular
    grid inside its rectangle (so that they rasterise to a solid block); the rest of the frame is invalid (0,0,0)."""
    rng = np.random.default_rng(seed)
    pts = []
    for z, n, (x0, x1), (y0, y1) in planes:
        nx = max(1, int(round(np.sqrt(n * (x1 - x0) / max(y1 - y0, 1e-9)))))
        ny = (n + nx - 1) // nx
        gx, gy = np.meshgrid(np.linspace(x0, x1, nx), np.linspace(y0, y1, ny))
        p = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, z)], 1)[:n]
        pts.append(p)
    if extra is not None:
        pts.append(np.asarray(extra, dtype=np.float64))
    p = np.concatenate(pts) if pts else np.zeros((0, 3))
    assert len(p) <= W * H
    out = np.zeros((W * H, 3), dtype=np.float32)
    idx = rng.permutation(W * H)[:len(p)]
    p = p.copy()
    p[:, 2] += Z_SHIFT                                  # world -> camera
    out[np.sort(idx)] = p.astype(np.float32)
    return out.reshape(H, W, 3)


Z_SHIFT = 0.5


def calibration(ssd):
    t = ssd.GeometricTransformation()                   # identity (transformation.h:51-55) ...
    t.constants.b[2] = -Z_SHIFT                         # ... with the camera half a metre below the world origin
    return t


def run(ssd, oracle, gpu_device, xyz):
    trans = calibration(ssd)
    cfg = ssd.default_config(W, H, max_frames_per_batch=1)
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
    dbg_res = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz)[0]
    det.close()
    return rep, dbg_res


GROUND = (0.005, 60000, (-0.55, 0.55), (0.15, 0.45))          # bins 10/11 with float rounding: a ground plateau
STEP1 = (0.1755, 30000, (-0.4, 0.4), (0.5, 0.75))
STEP2 = (0.3455, 25000, (-0.4, 0.4), (0.8, 1.05))


def test_plain_two_steps_hand_built(ssd, oracle, gpu_device):
    rep, res = run(ssd, oracle, gpu_device, cloud([GROUND, STEP1, STEP2]))
    assert res.n_plateaus == 3 and res.n_steps >= 2


def test_q1_flat_top_peak_reports_its_last_index(ssd, oracle, gpu_device):
    """Two neighbouring bins with exactly equal counts: findPeaks pushes the LAST flat index (pointcloud.cpp:227-238)."""
    half = 9000
    planes = [GROUND, (0.2055, half, (-0.4, 0.4), (0.5, 0.7)), (0.2155, half, (-0.4, 0.4), (0.7, 0.9)),
              (0.1955, 1000, (-0.4, 0.4), (0.95, 1.0)), (0.2255, 1000, (-0.4, 0.4), (1.0, 1.05))]
    rep, res = run(ssd, oracle, gpu_device, cloud(planes))
    hist = list(res.hist[:res.n_bins])
    assert hist[30] == hist[31] == half
    assert 31 in list(res.peaks[:res.n_peaks]) and 30 not in list(res.peaks[:res.n_peaks])


def test_q2_threshold_2000_is_absolute_and_sharpness_rule(ssd, oracle, gpu_device):
    """A peak of 1999 points is dropped, one of 2000 kept; a blunt peak ((2p-l-r)*2 <= p) is dropped (pointcloud.cpp:250-253)."""
    planes = [GROUND,
              (0.2055, 1999, (-0.3, 0.3), (0.5, 0.6)),                     # bin 30: below the threshold
              (0.4055, 2000, (-0.3, 0.3), (0.65, 0.75)),                   # bin 50: exactly at it
              (0.6055, 4000, (-0.3, 0.3), (0.8, 0.9)), (0.5955, 3000, (-0.3, 0.3), (0.9, 1.0)), (0.6155, 3001, (-0.3, 0.3), (1.0, 1.1))]
    rep, res = run(ssd, oracle, gpu_device, cloud(planes))
    peaks = list(res.peaks[:res.n_peaks])
    assert 30 not in peaks and 50 in peaks
    assert 70 in list(res.peaks_raw[:res.n_peaks_raw]) and 70 not in peaks      # (8000-3000-3001)*2 = 3998 <= 4000


def test_q3_pair_selection_tie_goes_up(ssd, oracle, gpu_device):
    """hist[h-1] == hist[h+1]: the strict '>' picks [h, h+1] (pointcloud.cpp:307-316)."""
    planes = [GROUND, (0.3055, 20000, (-0.4, 0.4), (0.5, 0.8)), (0.2955, 3000, (-0.4, 0.4), (0.85, 0.9)), (0.3155, 3000, (-0.4, 0.4), (0.95, 1.0))]
    rep, res = run(ssd, oracle, gpu_device, cloud(planes))
    k = [i for i in range(res.n_plateaus) if res.plateaus[i].peak_bin == 40]
    assert k and (res.plateaus[k[0]].bin_lo, res.plateaus[k[0]].bin_hi) == (40, 41)


def test_q4_pair_starting_at_bin_zero_swallows_everything(ssd, oracle, gpu_device):
    """Peak in bin 1 with hist[0] > hist[2]: Height_t(heightMin - 1) wraps, every point goes to the remainder and all
    later plateaus are empty (pointcloud.cpp:324, 337-343) -> no steps at all."""
    planes = [(-0.0945, 9000, (-0.5, 0.5), (0.15, 0.3)),       # bin 0
              (-0.0845, 20000, (-0.5, 0.5), (0.3, 0.5)),       # bin 1: the peak
              (-0.0745, 3000, (-0.5, 0.5), (0.5, 0.55)),       # bin 2
              STEP1, STEP2]
    rep, res = run(ssd, oracle, gpu_device, cloud(planes))
    assert res.plateaus[0].peak_bin == 1 and res.plateaus[0].bin_lo == 0
    assert all(res.plateaus[i].n_points == 0 for i in range(res.n_plateaus))
    assert res.n_steps == 0 and rep["line"] == '["stairs",["stairSteps",0]]'


def test_overlapping_pairs_later_plateau_gets_what_is_left(ssd, oracle, gpu_device):
    """Peaks two bins apart share the bin between them: the lower plateau takes it (extractPlateauPoints consumes in order)."""
    planes = [GROUND, (0.2055, 12000, (-0.4, 0.4), (0.5, 0.7)), (0.2155, 6000, (-0.4, 0.4), (0.7, 0.8)), (0.2255, 12001, (-0.4, 0.4), (0.8, 1.0)),
              (0.1955, 100, (-0.4, 0.4), (1.05, 1.1)), (0.2355, 100, (-0.4, 0.4), (1.1, 1.15))]
    rep, res = run(ssd, oracle, gpu_device, cloud(planes))
    pl = {res.plateaus[i].peak_bin: res.plateaus[i] for i in range(res.n_plateaus)}
    assert 30 in pl and 32 in pl
    assert (pl[30].bin_lo, pl[30].bin_hi) == (30, 31) and (pl[32].bin_lo, pl[32].bin_hi) == (31, 32)
    assert pl[30].n_points == 18000 and pl[32].n_points == 12001


def test_non_finite_and_huge_coordinates_are_dropped_like_the_reference(ssd, oracle, gpu_device):
    """NaN / inf / 1e30 coordinates fail the same strict compares in both implementations."""
    bad = np.array([[np.nan, 0.5, 0.2], [0.1, np.nan, 0.2], [0.1, 0.5, np.nan], [np.inf, 0.5, 0.2], [0.1, -np.inf, 0.2],
                    [0.1, 0.5, np.inf], [1e30, 0.5, 0.2], [0.1, 0.5, 1e30], [0.1, 0.5, -Z_SHIFT], [-0.0, 0.5, 3e-45 - Z_SHIFT], [0.1, 0.5, -0.7]] * 50)
    rep, res = run(ssd, oracle, gpu_device, cloud([GROUND, STEP1, STEP2], extra=bad))
    assert res.n_nonzero > res.n_inrange


def test_points_exactly_on_the_range_limits_are_outside(ssd, oracle, gpu_device):
    """All six limits are strict (configuration.h:44-46, pointcloud.cpp:157-162): 0.1 m and 1.3 m in y, 1.1 in z as
    float32 are a hair beyond/below the double limits; neighbours one float ulp inside are kept."""
    f = np.float32
    edge = []
    for v in (f(-0.6), f(0.6), np.nextafter(f(-0.6), f(0)), np.nextafter(f(0.6), f(0))):
        edge.append([v, 0.5, 0.2])
    for v in (f(0.1), f(1.3), np.nextafter(f(0.1), f(1)), np.nextafter(f(1.3), f(0))):
        edge.append([0.0, v, 0.2])
    for v in (1.1, 1.1 - 1e-7, -0.1, -0.1 + 1e-7, -0.1 - 1e-7):
        edge.append([0.0, 0.5, v])
    rep, res = run(ssd, oracle, gpu_device, cloud([GROUND, STEP1], extra=np.array(edge * 20, dtype=np.float64)))
    assert res.n_inrange < res.n_nonzero


def test_more_step_plateaus_than_image_slots_is_flagged(ssd, gpu_device):
    """The reference has no limit on plateaus; the workspace holds max_step_plateaus images per frame: a frame with more
    is processed up to the limit and flagged SSD_ST_OVERFLOW (never silently)."""
    planes = [GROUND] + [(0.1055 + 0.05 * k, 2500, (-0.4, 0.0), (0.5, 1.2)) for k in range(19)]
    xyz = cloud(planes)
    trans = calibration(ssd)
    cfg = ssd.default_config(W, H, max_frames_per_batch=1)
    det = ssd.Detector(cfg, trans, gpu_device)
    det.set_debug(True)
    fr = det.process_host(xyz)[0]
    dbg = det.debug(0)
    det.close()
    assert dbg.n_plateaus == 20 and (fr.status & ssd.ST_OVERFLOW)
    cfg2 = ssd.default_config(W, H, max_frames_per_batch=1)
    det = ssd.Detector(cfg2, trans, gpu_device)
    ok = det.process_host(cloud(planes[:17]))[0]                # ground + 16 step plateaus fit
    det.close()
    assert not (ok.status & ssd.ST_OVERFLOW)


@pytest.mark.parametrize("res", [(64, 48), (160, 120), (200, 150), (333, 257), (802, 602), (1282, 722), (2048, 1536)])
def test_resolutions_off_the_beaten_path(ssd, oracle, gpu_device, res):
    """Resolutions no camera mode has — tiny, odd, not a multiple of 4 / 64 / 50, larger than FHD: the scan-column strips of
    the ground raster (W/2 mod 50), the cell columns, the unaligned 12-byte loads and the bit images' ragged last words all
    depend on them.  A 3-step scene and a bare-ground scene each; every intermediate where the oracle is quick enough, results
    always; single call and a batch of three."""
    W, H = res
    for n_steps in (3, 0):
        sc = ssd.make_scene(W, H, n_steps=n_steps, seed=4200 + W + n_steps, sigma=0.0015)
        trans = ssd.transformation_for_scene(sc)
        cfg = ssd.default_config(W, H, max_frames_per_batch=3)
        xyz = ssd.synth_host([sc])[0]
        det = ssd.Detector(cfg, trans, gpu_device)
        rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=W * H <= 1300000)
        assert rep.get("max_height_err", 0.0) <= parity.TOL_HEIGHT and rep.get("max_corner_err", 0.0) == 0.0
        three = np.stack([xyz, xyz[::-1].copy() * 0.0, xyz])             # the frame, an all-invalid frame, the frame again
        res3 = det.process_host(three)
        assert bytes(res3[0]) == bytes(res3[2]) and res3[1].n_steps == 0
        parity.check_results_only(ssd, oracle, cfg, trans.constants, xyz, res3[0])
        if W % 4 == 0:
            # the same scene as 16-bit depth (round 4: the row of a pixel by a multiply-high above 128 columns, by a division
            # below; the x-map as one 16-byte load): results = the oracle's on the deprojected frame
            intr = ssd.intrinsics_for_scene(sc)
            depth = ssd.synth_depth_host([sc])[0]
            det.set_intrinsics(intr)
            got = det.process_depth_host(np.stack([depth, depth]))
            assert bytes(got[0]) == bytes(got[1])
            parity.check_results_only(ssd, oracle, cfg, trans.constants, oracle.deproject(intr, depth), got[0])
        det.close()


def _ring_cloud(copies=3):
    """A 512 x 512 frame over a measuring range of exactly one metre each way (pixel borders = multiples of 1/512 m, exact in
    float32): a dense ground plane and ONE step plateau that consists only of the outline of a rectangle, its points lying
    exactly on pixel corners.  The outline the reference finds passes through those very coordinates, and its point test is
    strict (quadrilateralTest.cpp:58-61, :145-166) — so the step is valid, convex, and accepts none of its own points."""
    n = 512
    px = lambda c: -0.5 + c / 512.0
    py = lambda r: 1.25 - r / 512.0
    gx, gy = np.meshgrid(np.arange(60, 450), np.arange(400, 500))
    pts = [np.stack([px(gx.ravel() + 0.5), py(gy.ravel() + 0.5), np.full(gx.size, 0.005)], 1)]
    r0, r1, c0, c1 = 200, 330, 100, 410
    ring = [(c, r) for c in range(c0, c1 + 1) for r in (r0, r1)] + [(c, r) for r in range(r0 + 1, r1) for c in (c0, c1)]
    ring = np.array(ring)
    for _ in range(copies):                              # 880 ring points x 3: above filterPeaks' 2000
        pts.append(np.stack([px(ring[:, 0]), py(ring[:, 1]), np.full(len(ring), 0.1755)], 1))
    p = np.concatenate(pts)
    p[:, 2] += Z_SHIFT
    out = np.zeros((n * n, 3), dtype=np.float32)
    out[:len(p)] = p.astype(np.float32)
    return out.reshape(n, n, 3)


def test_step_whose_quadrilateral_accepts_no_point_reads_minus_nan(ssd, oracle, gpu_device):
    """VERDICT round 3, item 3.  calcAverageZ divides 0.0 by 0 for a quadrilateral without points (pointcloud.cpp:574-581):
    on the reference's x86 that is the default NaN with the sign bit set, and the line reads "-nan" (stairs.cpp:43).  The
    GPU's own 0.0 / 0.0 is the positive NaN ("nan"): k_final states the empty case instead of dividing.  A hand-built cloud
    reaches it: every intermediate, the result's sign bit and the line's bytes against the oracle."""
    xyz = _ring_cloud()
    trans = calibration(ssd)
    cfg = ssd.default_config(512, 512, max_frames_per_batch=1)
    cfg.x_min, cfg.x_max, cfg.y_min, cfg.y_max = -0.5, 0.5, 0.25, 1.25
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
    fr = det.process_host(xyz)[0]
    det.close()
    assert rep["n_steps"] == 2 and fr.n_steps == 2
    h = fr.steps[1].height
    assert np.isnan(h) and np.signbit(h), "the empty quadrilateral's mean must be x86's default NaN (sign bit set)"
    n, steps, status = oracle.process_lean(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz)
    assert n == 2 and np.isnan(steps[1][0]) and np.signbit(steps[1][0])
    line = ssd.Stairs(fr).serialize()
    assert line == oracle.serialize(steps) and '["height",-nan]' in line and line.count("nan") == 1


def test_ground_quadrilateral_without_points_reads_minus_nan(ssd, oracle, gpu_device):
    """The same for the ground's sum (calcGround, pointcloud.cpp:528-547), which no cloud was found to empty: the state is
    rewritten between k_inquad and k_final (test hook) as if the ground quadrilateral had accepted nothing.  The frame's other
    numbers must not move, the ground's height must read -nan."""
    W_, H_ = 640, 480
    sc = ssd.make_scene(W_, H_, n_steps=3, seed=515, sigma=0.001)
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(W_, H_, max_frames_per_batch=1)
    xyz = ssd.synth_host([sc])[0]
    det = ssd.Detector(cfg, trans, gpu_device)
    good = det.process_host(xyz)[0]
    assert good.n_steps == 4
    buf = ssd.DeviceBuffer(W_ * H_ * 12, gpu_device)
    buf.upload(xyz)
    det.enqueue(buf.ptr, 1, stages=ssd.STAGE_ALL & ~ssd.STAGE_FINAL)
    det.empty_quadrilateral(0, -1)
    det.enqueue(buf.ptr, 1, stages=ssd.STAGE_FINAL)
    fr = det.fetch_list(1)[0]
    det.close()
    buf.free()
    assert fr.n_steps == 4 and np.isnan(fr.steps[0].height) and np.signbit(fr.steps[0].height)
    assert [list(fr.steps[i].quad) for i in range(4)] == [list(good.steps[i].quad) for i in range(4)]
    assert [fr.steps[i].height for i in range(1, 4)] == [good.steps[i].height for i in range(1, 4)]
    want = ssd.Stairs(good).serialize()
    cut = want.index('["height",') + len('["height",')
    assert ssd.Stairs(fr).serialize() == want[:cut] + "-nan" + want[want.index("]", cut):]


def _band_cloud(cfg, a, b, rng, n_per):
    """camera-frame float points whose world x or y lies on / next to a limit of the measuring range (see the test below)"""
    inv = np.linalg.inv(a)
    deltas = np.concatenate([[0.0], *[[d, -d] for d in (2.2e-16, 1e-15, 1e-12, 1e-10, 1e-9, 1e-8, 3e-8, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 4.5e-5, 5.5e-5, 7e-5, 1e-4)]])
    pts = []
    for lim, axis in ((cfg.x_min, 0), (cfg.x_max, 0), (cfg.y_min, 1), (cfg.y_max, 1)):
        for d in list(deltas) + [None]:
            w = np.empty((n_per, 3))
            w[:, 0] = rng.uniform(cfg.x_min + 0.05, cfg.x_max - 0.05, n_per)
            w[:, 1] = rng.uniform(cfg.y_min + 0.05, cfg.y_max - 0.05, n_per)
            w[:, 2] = rng.choice([0.0034, 0.1712, 0.3391], n_per) + rng.normal(0.0, 0.0007, n_per)
            w[:, axis] = lim + (rng.uniform(-2e-4, 2e-4, n_per) if d is None else d)     # None: anywhere around the band's own edges
            pts.append((w - b) @ inv.T)
    # corners of the range: both coordinates in the band at once
    for xl in (cfg.x_min, cfg.x_max):
        for yl in (cfg.y_min, cfg.y_max):
            w = np.stack([xl + rng.normal(0, 2e-7, n_per), yl + rng.normal(0, 2e-7, n_per), np.full(n_per, 0.1712)], 1)
            pts.append((w - b) @ inv.T)
    return np.concatenate(pts).astype(np.float32)


def test_points_in_the_prefilters_band_take_the_doubles(ssd, oracle, gpu_device):
    """K1 decides the x / y range test in single precision first (csrc/ssd_prexy.h) and hands the band around the four limits -
    and inputs beyond 64 m - to the reference's doubles.  A pitched, rolled, yawed camera and a cloud made for that band:
    world points whose x or y is a limit plus or minus nothing, a few double ulps, 1e-12 .. 1e-4 m (the band's own edges lie
    at +- 4.5e-5 m here), taken back to camera coordinates and rounded to float (which scatters them by ~1e-7 m to either
    side of the limit), on two plateaus' heights; far points (70 .. 5000 m out) along all axes; everything else a plain
    two-step cloud.  Then the same rotation seen from 40 m away, where single precision is off by 4e-7 of the range instead
    of 6e-8 (a band of one float ulp gets a thousand of these points wrong; the bound must hold them all).  Histogram,
    counts, images, results: the oracle's, bit for bit - one point called wrongly moves a bin's count."""
    sc = ssd.make_scene(W, H, n_steps=2, seed=77, pitch_deg=47.0, roll_deg=3.5, yaw_deg=12.0, sigma=0.001)
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(W, H, max_frames_per_batch=1)
    a = np.array(list(trans.constants.a), dtype=np.float64).reshape(3, 3)
    b = np.array(list(trans.constants.b), dtype=np.float64)
    rng = np.random.default_rng(5)
    xyz = ssd.synth_host([sc])[0].reshape(-1, 3).copy()
    band = _band_cloud(cfg, a, b, rng, 300)
    far = []
    for r in (70.0, 300.0, 5000.0):
        for ax in range(3):
            for sgn in (1.0, -1.0):
                v = np.zeros(3); v[ax] = sgn * r
                far.append(v + rng.normal(0, 0.3, (40, 3)))
    far = np.concatenate(far).astype(np.float32)
    extra = np.concatenate([band, far])
    assert len(extra) < W * H // 2
    idx = np.sort(rng.permutation(W * H)[:len(extra)])
    xyz[idx] = extra
    xyz = xyz.reshape(H, W, 3)
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
    ref = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants), xyz)[0]
    assert ref.n_inrange < ref.n_nonzero and rep["n_steps"] >= 1
    # the same cloud through the single pass (forced: K1 then also rasters the candidate bins' points itself)
    det.single_pass(1)
    rep1 = parity.check_frame(ssd, oracle, det, cfg, trans.constants, xyz, images=True)
    assert det.single_pass_stats(1)["ran"] and rep1["line"] == rep["line"]
    det.close()
    # 40 m away: a calibration with the same rotation whose camera stands far off (inputs still below the pre-filter's 64 m)
    far_trans = ssd.GeometricTransformation()
    for i in range(9):
        far_trans.constants.a[i] = trans.constants.a[i]
    b_far = b + a @ np.array([3.0, -2.0, -40.0])
    for i in range(3):
        far_trans.constants.b[i] = b_far[i]
    band_far = _band_cloud(cfg, a, b_far, rng, 300)
    assert 30.0 < np.abs(band_far).max() < 60.0
    cloud_far = np.zeros((W * H, 3), dtype=np.float32)
    cloud_far[np.sort(rng.permutation(W * H)[:len(band_far)])] = band_far
    det = ssd.Detector(cfg, far_trans, gpu_device)
    parity.check_frame(ssd, oracle, det, cfg, far_trans.constants, cloud_far.reshape(H, W, 3), images=True)
    det.close()


def _bin_edge_cloud(cfg, a, b, rng, n_per):
    """camera-frame float points whose world z lies on / next to EVERY edge of a height bin - both limits of the z range among them -
    plus or minus nothing, a few double ulps, 1e-14 .. 1e-3 of a bin and the single-precision band's own edges (about 1e-4 of a bin at
    these distances); x / y anywhere in range (a few on its limits as well: both bands at once)"""
    inv = np.linalg.inv(a)
    n_bins = int((cfg.z_max - cfg.z_min) / cfg.height_interval) + 1
    deltas = np.concatenate([[0.0], *[[d, -d] for d in (1e-14, 1e-12, 1e-10, 1e-8, 1e-7, 1e-6, 1e-5, 3e-5, 6e-5, 9e-5, 1.2e-4, 1.5e-4, 2e-4, 4e-4, 1e-3)]])
    pts = []
    for edge in range(0, n_bins + 1):
        w = np.empty((n_per, 3))
        w[:, 0] = rng.uniform(cfg.x_min + 0.01, cfg.x_max - 0.01, n_per)
        w[:, 1] = rng.uniform(cfg.y_min + 0.01, cfg.y_max - 0.01, n_per)
        w[:, 2] = cfg.z_min + (edge + rng.choice(deltas, n_per)) * cfg.height_interval
        k = n_per // 8                                           # an eighth of them on an x or y limit too
        w[:k, 0] = rng.choice([cfg.x_min, cfg.x_max], k) + rng.normal(0, 3e-5, k)
        pts.append((w - b) @ inv.T)
    return np.concatenate(pts).astype(np.float32)


def _pixel_edge_cloud(cfg, a, b, rng, heights, n_per, width, height):
    """camera-frame float points at the heights of the treads (so that they fall into bins K1 rasters itself in the single pass) whose
    top-down pixel coordinate lies on / next to a pixel edge in x or in y - the image's own borders among them -, placed beside the
    staircase (|x| > 0.45 m) where no other point of those bins lights the pixels: a point rastered one pixel off shows in the image"""
    inv = np.linalg.inv(a)
    x_to_img, y_to_img = width / (cfg.x_max - cfg.x_min), height / (cfg.y_max - cfg.y_min)
    deltas = np.concatenate([[0.0], *[[d, -d] for d in (1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 3e-4, 6e-4, 1e-3, 2e-3, 5e-3)]])
    pts = []
    for z in heights:
        for axis in (0, 1):
            w = np.empty((n_per, 3))
            side = rng.choice([-1.0, 1.0], n_per)
            w[:, 0] = side * rng.uniform(0.46, 0.59, n_per)
            w[:, 1] = rng.uniform(cfg.y_min + 0.02, cfg.y_max - 0.02, n_per)
            w[:, 2] = z + rng.normal(0.0, 0.0004, n_per)
            d = rng.choice(deltas, n_per)
            if axis == 0:
                col = np.floor((w[:, 0] - cfg.x_min) * x_to_img)
                col[: n_per // 10] = rng.choice([0.0, float(width)], n_per // 10)        # the image's left and right border
                w[:, 0] = cfg.x_min + (col + d) / x_to_img
            else:
                row = np.floor((cfg.y_max - w[:, 1]) * y_to_img)
                row[: n_per // 10] = rng.choice([0.0, float(height)], n_per // 10)       # top and bottom border
                w[:, 1] = cfg.y_max - (row + d) / y_to_img
            pts.append((w - b) @ inv.T)
    return np.concatenate(pts).astype(np.float32)


@pytest.mark.parametrize("far_camera", [False, True])
def test_points_on_bin_edges_and_pixel_edges_take_the_doubles(ssd, oracle, gpu_device, far_camera):
    """Round 6: K1 takes the height bin, the z-range test and (single pass) the pixel of a candidate point from single precision
    first, with a bound that follows the point's magnitude (csrc/ssd_prexy.h: make_pre_z, make_pre_pixel), and hands the points
    within that bound of a bin edge / pixel edge to the reference's doubles.  A cloud made for those bands: world z on every bin
    edge (the range's limits are the first and the last) and world x / y on pixel edges beside the staircase, plus or minus nothing
    .. 1e-3 of a bin / 5e-3 of a pixel; rounding to camera floats scatters them by ~1e-5 of a bin to either side.  Histogram (one
    point in the wrong bin moves two counts), counts, raw images (one pixel off shows: nothing else lights those pixels), results:
    the oracle's, bit for bit, two passes and single pass.  With the camera 25 m away single precision is ten times as coarse and
    the band ten times as wide.  (A build with -DSSD_SABOTAGE_PRE=1 / =2 - the bands not handed over - fails this test:
    profiles/r06_prefilter_sabotage.txt.)"""
    sc = ssd.make_scene(W, H, n_steps=2, seed=78, pitch_deg=46.0, roll_deg=-2.5, yaw_deg=-9.0, sigma=0.001)
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(W, H, max_frames_per_batch=1)
    a = np.array(list(trans.constants.a), dtype=np.float64).reshape(3, 3)
    b = np.array(list(trans.constants.b), dtype=np.float64)
    rng = np.random.default_rng(6)
    xyz = ssd.synth_host([sc])[0].reshape(-1, 3).copy()
    if far_camera:
        # the same rotation seen from 25 m away: the staircase's own points moved along, so that the frame still has its treads
        shift = np.array([2.0, -1.5, -25.0])
        use = ssd.GeometricTransformation()
        for i in range(9):
            use.constants.a[i] = trans.constants.a[i]
        b_use = b + a @ shift
        for i in range(3):
            use.constants.b[i] = b_use[i]
        for name in ("r2", "t2"):
            for i in range(len(getattr(trans.constants, name))):
                getattr(use.constants, name)[i] = getattr(trans.constants, name)[i]
        use.constants.world_z = trans.constants.world_z
        valid = xyz[:, 2] > 0
        xyz[valid] = (xyz[valid].astype(np.float64) - shift).astype(np.float32)       # world = a (p - shift) + b_use = a p_old + b
        xyz[valid & ~(xyz[:, 2] > 0)] = 0.0
    else:
        use, b_use = trans, b
    ref0 = oracle.process(ob.to_oracle_config(cfg), ob.to_oracle_calibration(use.constants), xyz.reshape(H, W, 3))[0]
    plateaus = [ref0.plateaus[i] for i in range(ref0.n_plateaus)]
    tread_heights = [cfg.z_min + (p.peak_bin + 0.5) * cfg.height_interval for p in plateaus if p.is_step][:2]
    assert len(tread_heights) >= 1
    edges = _bin_edge_cloud(cfg, a, b_use, rng, 400)
    pixels = _pixel_edge_cloud(cfg, a, b_use, rng, tread_heights, 1500, W, H)
    extra = np.concatenate([edges, pixels])
    assert len(extra) < W * H // 4
    if far_camera:
        assert 20.0 < np.abs(extra).max() < 40.0
    idx = np.sort(rng.permutation(W * H)[:len(extra)])
    xyz[idx] = extra
    xyz = xyz.reshape(H, W, 3)
    det = ssd.Detector(cfg, use, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, use.constants, xyz, images=True)
    det.single_pass(1)
    rep1 = parity.check_frame(ssd, oracle, det, cfg, use.constants, xyz, images=True)
    assert det.single_pass_stats(1)["ran"] and rep1["line"] == rep["line"] and rep["n_steps"] >= 1
    det.close()


def _world(cfg, a, b, p32):
    """the reference's rows on camera floats (transformation.h:59-64, the operations in its order), the range test, bin and pixel"""
    x, y, z = (p32[:, k].astype(np.float64) for k in range(3))
    wx = ((a[0, 0] * x + a[0, 1] * y) + a[0, 2] * z) + b[0]
    wy = ((a[1, 0] * x + a[1, 1] * y) + a[1, 2] * z) + b[1]
    wz = ((a[2, 0] * x + a[2, 1] * y) + a[2, 2] * z) + b[2]
    ok = (p32[:, 2] > 0) & (wx > cfg.x_min) & (wx < cfg.x_max) & (wy > cfg.y_min) & (wy < cfg.y_max) & (wz > cfg.z_min) & (wz < cfg.z_max)
    recip = 1.0 / cfg.height_interval
    with np.errstate(invalid="ignore"):
        hbin = np.where(ok, (wz - cfg.z_min) * recip, 0.0).astype(np.int64)
        ix = np.where(ok, (wx - cfg.x_min) * (W / (cfg.x_max - cfg.x_min)), 0.0).astype(np.int64)
        iy = np.where(ok, (cfg.y_max - wy) * (H / (cfg.y_max - cfg.y_min)), 0.0).astype(np.int64)
    return wx, wy, wz, ok, hbin, ix, iy


def _snap_onto_quad_edges(cfg, a, b, xyz, surfaces, rng, per_edge=400):
    """Moves points of each surface (quadrilateral as 4 x (x, y); its height bins lo .. hi) that lie within 1.2 mm of one of its edges
    ONTO that edge (or a chosen distance from nothing to 10 um off it) - by less than a pixel and only if the moved point keeps its bin
    and its pixel, so that histogram and rasters, hence plateaus, outlines and quadrilaterals, stay what they were.  Rounding to camera
    floats scatters the moved points by ~1e-7 m to either side of the edge.  Returns (the cloud, distances of the moved points from
    their edge)."""
    p = xyz.reshape(-1, 3).copy()
    wx, wy, wz, ok, hbin, ix, iy = _world(cfg, a, b, p)
    inv = np.linalg.inv(a)
    offsets = np.array([0.0, 0.0, 1e-9, -1e-9, 1e-7, -1e-7, 5e-7, -5e-7, 1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5])
    moved = np.zeros(len(p), dtype=bool)
    dists = []
    for quad, lo, hi in surfaces:
        q = np.asarray(quad, dtype=np.float64).reshape(4, 2)
        mine = ok & (hbin >= lo) & (hbin <= hi) & ~moved
        for i0, i1 in ((0, 1), (1, 3), (3, 2), (2, 0)):
            p0, along = q[i0], q[i1] - q[i0]
            n = np.array([-along[1], along[0]]) / np.hypot(*along)
            t = ((wx - p0[0]) * along[0] + (wy - p0[1]) * along[1]) / (along @ along)
            dist = (wx - p0[0]) * n[0] + (wy - p0[1]) * n[1]
            cand = np.flatnonzero(mine & ~moved & (t > 0.02) & (t < 0.98) & (np.abs(dist) < 1.2e-3))
            cand = rng.permutation(cand)[:per_edge]
            if len(cand) == 0:
                continue
            off = rng.choice(offsets, len(cand))
            w = np.stack([wx[cand] - (dist[cand] - off) * n[0], wy[cand] - (dist[cand] - off) * n[1], wz[cand]], 1)
            new = ((w - b) @ inv.T).astype(np.float32)
            nwx, nwy, nwz, nok, nbin, nix, niy = _world(cfg, a, b, new)
            keep = nok & (nbin == hbin[cand]) & (nix == ix[cand]) & (niy == iy[cand])
            p[cand[keep]] = new[keep]
            moved[cand[keep]] = True
            dists.append(np.abs((nwx[keep] - p0[0]) * n[0] + (nwy[keep] - p0[1]) * n[1]))
    return p.reshape(xyz.shape), np.concatenate(dists) if dists else np.zeros(0)


@pytest.mark.parametrize("yaw_deg", [-9.0, 14.0])
def test_points_on_the_edges_of_the_quadrilaterals_take_the_doubles(ssd, oracle, gpu_device, yaw_deg):
    """Round 6: k_inquad asks four single-precision half-planes about a point first and the reference's QuadrilateralTest, in doubles,
    only for the points within a bound (some micrometres) of an edge (csrc/ssd_quadtest.h: build_quad_edges).  A frame made for
    that band: points of the treads and of the ground moved onto the edges of their own quadrilaterals - within their pixel and bin,
    so that the quadrilaterals stay where they are (checked) - hundreds of them within a micrometre of an edge, on either side.
    Which side decides whether the point counts: the numbers of points in the quadrilaterals, their mean heights, the ground image
    and the result are the oracle's.  (A build with -DSSD_SABOTAGE_PRE=4 - the band not handed to the doubles - fails this test:
    profiles/r06_prefilter_sabotage.txt.)"""
    sc = ssd.make_scene(W, H, n_steps=3, seed=79, pitch_deg=44.0, roll_deg=1.5, yaw_deg=yaw_deg, sigma=0.001)
    trans = ssd.transformation_for_scene(sc)
    cfg = ssd.default_config(W, H, max_frames_per_batch=1)
    a = np.array(list(trans.constants.a), dtype=np.float64).reshape(3, 3)
    b = np.array(list(trans.constants.b), dtype=np.float64)
    rng = np.random.default_rng(11)
    xyz = ssd.synth_host([sc])[0].reshape(H, W, 3).copy()
    ocfg, ocal = ob.to_oracle_config(cfg), ob.to_oracle_calibration(trans.constants)
    ref0 = oracle.process(ocfg, ocal, xyz)[0]
    assert ref0.n_steps >= 2 and ref0.first_valid_ind >= 0 and ref0.ground_ind >= 0 and not (ref0.status & ob.ST_THROW)
    surfaces = []
    for k in range(ref0.n_plateaus):
        pl = ref0.plateaus[k]
        if pl.is_step and pl.valid:
            surfaces.append((list(pl.quad_world), pl.bin_lo, pl.bin_hi))
    g = ref0.plateaus[ref0.ground_ind]
    surfaces.append((list(ref0.ground_quad_world), g.bin_lo, g.bin_hi))
    made, dist = _snap_onto_quad_edges(cfg, a, b, xyz, surfaces, rng)
    assert len(dist) > 1500 and (dist < 1e-6).sum() > 400 and (dist < 2e-7).sum() > 100, (len(dist), (dist < 1e-6).sum())
    ref1 = oracle.process(ocfg, ocal, made)[0]
    # nothing upstream of the quadrilateral tests has moved
    assert list(ref1.hist) == list(ref0.hist) and list(ref1.ground_quad_world) == list(ref0.ground_quad_world)
    for k in range(ref0.n_plateaus):
        assert list(ref1.plateaus[k].quad_world) == list(ref0.plateaus[k].quad_world)
    # ... and the points on the edges did change what is counted (so the test sees the band)
    counts0 = [ref0.plateaus[k].n_in_quad for k in range(ref0.n_plateaus)] + [ref0.ground_n_in_quad]
    counts1 = [ref1.plateaus[k].n_in_quad for k in range(ref1.n_plateaus)] + [ref1.ground_n_in_quad]
    assert counts0 != counts1
    det = ssd.Detector(cfg, trans, gpu_device)
    rep = parity.check_frame(ssd, oracle, det, cfg, trans.constants, made, images=True)
    det.single_pass(1)
    rep1 = parity.check_frame(ssd, oracle, det, cfg, trans.constants, made, images=True)
    assert rep1["line"] == rep["line"] and rep["n_steps"] >= 2
    det.close()
