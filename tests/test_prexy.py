"""K1's single-precision pre-filter of the x / y range test (csrc/ssd_prexy.h: make_pre_xy) on the CPU: the bound it derives
against double precision, on random calibrations and measuring ranges and on points made for the band around the limits.

The kernel computes d = fma(c0, x, fma(c1, y, fma(c2, z, c3))) per row in single precision and calls a point "inside" when
max(|dx|, |dy|) < lo and "outside" when it is > hi; everything else goes through the reference's doubles.  Here the same chain is
evaluated with numpy (each FMA as an exact float64 product-and-sum rounded once to float32: the product of two float32 values
and the sum fit float64 to well below the bound's slack) and held against the exact value in float64 / longdouble."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ssd = importlib.import_module("stair-step-detector_amd")


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _chain(c_row, p):
    """the kernel's three FMAs of one row on float32 inputs p [n, 3]; c_row = (c0, c1, c2, c3) float32"""
    f = np.float64
    r = (f(c_row[2]) * p[:, 2].astype(f) + f(c_row[3])).astype(np.float32)
    r = (f(c_row[1]) * p[:, 1].astype(f) + r.astype(f)).astype(np.float32)
    return (f(c_row[0]) * p[:, 0].astype(f) + r.astype(f)).astype(np.float32)


def _exact(a, b, lo, hi, p):
    """D = ((a . p + b) - lo) / (hi - lo) - 1/2 in extended precision"""
    L = np.longdouble
    w = a[0].astype(L) * p[:, 0].astype(L) + a[1].astype(L) * p[:, 1].astype(L) + a[2].astype(L) * p[:, 2].astype(L) + L(b)
    return ((w - L(lo)) / (L(hi) - L(lo)) - L(0.5)).astype(np.float64), w.astype(np.float64)


@pytest.mark.parametrize("seed", range(12))
def test_the_bound_holds_on_random_calibrations_and_ranges(seed):
    rng = np.random.default_rng(1000 + seed)
    a = _rotation(rng)
    cam = rng.uniform(-1.0, 1.0, 3) * (2.0 if seed % 3 else 45.0)          # every third camera stands far off: large inputs in range
    b = -a @ cam                                                          # world = a (p - cam)
    x_min, y_min = rng.uniform(-1.5, -0.3), rng.uniform(0.0, 0.5)
    x_max, y_max = x_min + rng.uniform(0.6, 2.5), y_min + rng.uniform(0.6, 2.5)
    z_min, z_max = -0.1, 1.1
    Q = ssd.prexy_host(x_min, x_max, y_min, y_max, z_min, z_max, a, b)
    e = 0.5 - float(Q["lo"])
    assert 0.0 < e < 0.01 and float(Q["hi"]) - 0.5 >= e * 0.99 and Q["max_input"] == 64.0
    # world points: in the band around each limit, at the corners, anywhere in and around the range; back to camera floats
    n = 40000
    w = np.stack([rng.uniform(x_min - 0.3, x_max + 0.3, n), rng.uniform(y_min - 0.3, y_max + 0.3, n), rng.uniform(z_min, z_max, n)], 1)
    k = n // 2
    axis = rng.integers(0, 2, k)
    lim = np.where(axis == 0, rng.choice([x_min, x_max], k), rng.choice([y_min, y_max], k))
    offs = rng.choice([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 5e-5, -5e-5, 2e-4, -2e-4], k)
    w[np.arange(k), axis] = lim + offs + rng.normal(0.0, 1e-7, k)
    p = ((w - b) @ np.linalg.inv(a).T).astype(np.float32)
    keep = np.abs(p).max(1) <= 64.0
    p = p[keep]
    c = Q["c"]
    dx, dy = _chain(c[:, 0], p), _chain(c[:, 1], p)
    Dx, wx = _exact(a[0], b[0], x_min, x_max, p)
    Dy, wy = _exact(a[1], b[1], y_min, y_max, p)
    err = max(np.abs(dx - Dx).max(), np.abs(dy - Dy).max())
    assert err <= e, (err, e)
    M = np.maximum(np.abs(dx), np.abs(dy))
    inside = (wx > x_min) & (wx < x_max) & (wy > y_min) & (wy < y_max)     # what the reference's doubles decide
    assert not np.any((M < Q["lo"]) & ~inside) and not np.any((M > Q["hi"]) & inside)
    band = (M >= Q["lo"]) & (M <= Q["hi"])
    assert 0 < band.sum() < len(p)                                         # some points in the band, and not all of them


def test_large_inputs_read_outside_when_the_magnitude_test_is_dropped():
    """check_input == 0: make_pre_xy showed that an input beyond 64 m whose z is in range reads max(|dx|, |dy|) > hi.  Points
    64 .. 10^6 m out along every direction whose world z IS in range: they must read "outside", as the doubles say."""
    rng = np.random.default_rng(7)
    seen_dropped = 0
    for trial in range(8):
        a = _rotation(rng)
        cam = rng.uniform(-1.5, 1.5, 3)
        b = -a @ cam
        Q = ssd.prexy_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, a, b)
        if Q["check_input"]:
            continue
        seen_dropped += 1
        n = 20000
        r = 10.0 ** rng.uniform(np.log10(64.0), 6.0, n)
        ang = rng.uniform(0, 2 * np.pi, n)
        w = np.stack([r * np.cos(ang), r * np.sin(ang), rng.uniform(-0.1, 1.1, n)], 1)        # far out in the world's x / y plane, z in range
        p = ((w - b) @ np.linalg.inv(a).T).astype(np.float32)
        p = p[np.abs(p).max(1) > 64.0]
        with np.errstate(over="ignore", invalid="ignore"):
            M = np.maximum(np.abs(_chain(Q["c"][:, 0], p)), np.abs(_chain(Q["c"][:, 1], p)))
        assert np.all(M > Q["hi"])
    assert seen_dropped >= 4
    # a camera 45 m away from the range, or a calibration that is no rotation: the test stays
    far = ssd.prexy_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, np.eye(3), np.array([45.0, 0.0, 0.0]))
    skew = ssd.prexy_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, np.diag([1.0, 1.0, 0.2]) + 0.4, np.zeros(3))
    assert far["check_input"] and skew["check_input"]


def test_a_calibration_single_precision_cannot_serve_sends_every_point_through_the_doubles():
    Q = ssd.prexy_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, np.eye(3) * 1e7, np.zeros(3))
    assert Q["lo"] < 0.0 and np.isinf(Q["hi"]) and Q["check_input"]


# ---- round 6: the z row / height bin and the candidates' pixel in single precision first (make_pre_z, make_pre_pixel) ----

def _fma32(a, x, c):
    """fl32(a * x + c) on float32 operands: the product and the sum are formed in float64 and rounded once (the product of two
    float32 values is exact in float64; the sum's own float64 rounding is 2^-29 of a float32 ulp)"""
    return (np.float64(a) * x.astype(np.float64) + np.asarray(c, dtype=np.float32).astype(np.float64)).astype(np.float32)


def _t_chain(zc, p):
    return _fma32(zc[0], p[:, 0], _fma32(zc[1], p[:, 1], _fma32(zc[2], p[:, 2], np.full(len(p), zc[3], np.float32))))


def _sure(t, M3, neg_k, h0):
    """the kernel's certainty test: |fract(t) - 1/2| < h0 + neg_k * M3, all in float32; False for NaNs"""
    with np.errstate(invalid="ignore", over="ignore"):
        g = ((t - np.floor(t)).astype(np.float32) - np.float32(0.5)).astype(np.float32)
        h = _fma32(neg_k, M3, np.full(len(t), h0, np.float32))
        return np.abs(g) < h


def _reference_z(a, b, z_min, z_max, recip, p):
    """pointcloud.cpp:150-178 / transformation.h:59-64 in the reference's doubles, operation by operation"""
    x, y, z = (p[:, i].astype(np.float64) for i in range(3))
    wz = ((a[2, 0] * x + a[2, 1] * y) + a[2, 2] * z) + b[2]
    inz = (wz > z_min) & (wz < z_max)
    with np.errstate(invalid="ignore"):
        hb = ((wz - z_min) * recip).astype(np.int64)
    return wz, inz, hb


@pytest.mark.parametrize("seed", range(12))
def test_z_bound_bin_and_range_on_random_calibrations(seed):
    rng = np.random.default_rng(2000 + seed)
    a = _rotation(rng)
    cam = rng.uniform(-1.0, 1.0, 3) * (2.0 if seed % 3 else 30.0)
    b = -a @ cam
    z_min = -0.1 if seed % 4 else rng.uniform(-0.3, 0.0)
    interval = 0.01 if seed % 2 else rng.uniform(0.005, 0.02)
    z_max = (1.1 if seed % 4 else z_min + 1.0) if seed % 5 else z_min + 100.5 * interval        # every fifth range ends in the middle of a bin
    Z = ssd.prez_host(-0.6, 0.6, 0.1, 1.3, z_min, z_max, a, b, height_interval=interval)
    recip = 1.0 / interval
    if seed % 5 == 0:
        assert Z["z_check_top"]                          # the range ends in the middle of a bin
    elif seed % 2 == 1:
        assert not Z["z_check_top"]                      # the default range and interval: 120 bins and an ulp
    n_bins = int((z_max - z_min) * recip) + 1
    # world points with z on every bin edge and on both limits, +- 0, a few double ulps, 1e-12 .. 1e-4 bins, and anywhere; back to camera floats
    n = 60000
    w = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(0.0, 1.5, n), rng.uniform(z_min - 0.05, z_max + 0.05, n)], 1)
    k = 2 * n // 3
    edge = rng.integers(0, n_bins + 1, k).astype(np.float64)
    edge[: k // 8] = 0.0
    edge[k // 8: k // 4] = (z_max - z_min) * recip
    offs = rng.choice([0.0, 1e-13, -1e-13, 1e-11, -1e-11, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 3e-4, -3e-4, 1e-3, -1e-3], k)
    w[:k, 2] = z_min + (edge + offs + rng.normal(0.0, 2e-6, k)) * interval
    p = ((w - b) @ np.linalg.inv(a).T).astype(np.float32)
    t = _t_chain(Z["zc"], p)
    M3 = np.abs(p).max(1).astype(np.float32)
    # (1) the distance from the exact value stays inside the bound the threshold is made of
    L = np.longdouble
    T = ((a[2, 0].astype(L) * p[:, 0].astype(L) + a[2, 1].astype(L) * p[:, 1].astype(L) + a[2, 2].astype(L) * p[:, 2].astype(L) + L(b[2])) - L(z_min)) * L(recip)
    e = 0.5 - (np.float64(Z["z_h0"]) + np.float64(Z["z_neg_k"]) * M3.astype(np.float64))
    assert np.all(np.abs(t.astype(np.float64) - T.astype(np.float64)) <= e)
    # (2) a sure point has the reference's bin and the reference's range decision
    sure = _sure(t, M3, Z["z_neg_k"], Z["z_h0"])
    if Z["z_check_top"]:
        h = _fma32(Z["z_neg_k"], M3, np.full(len(t), Z["z_h0"], np.float32))
        sure &= np.abs((t - Z["z_top"]).astype(np.float32)) > (np.float32(0.5) - h).astype(np.float32)
    wz, inz, hb = _reference_z(a, b, z_min, z_max, recip, p)
    inzf = (t >= 0.0) & (t < Z["z_top"]) & ~np.signbit(t)
    assert np.array_equal(inzf[sure], inz[sure])
    both = sure & inz
    assert np.array_equal(np.floor(t[both]).astype(np.int64), hb[both])
    assert hb[both].min() >= 0 and hb[both].max() < n_bins
    # the band is thin, and it is there
    frac_unsure = 1.0 - sure[k:].mean()
    assert 0 < (~sure).sum() and frac_unsure < (0.02 if np.abs(cam).max() > 3 else 0.002), frac_unsure


def test_z_test_sends_nans_infinities_and_huge_inputs_through_the_doubles():
    a = _rotation(np.random.default_rng(5))
    b = -a @ np.array([0.1, -0.2, 1.0])
    Z = ssd.prez_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, a, b)
    big = np.float32(3e38)
    p = np.array([[np.nan, 1, 1], [1, np.nan, 1], [1, 1, np.nan], [np.inf, 1, 1], [1, -np.inf, 1], [1, 1, np.inf], [big, -big, 1], [big, big, big],
                  [1e30, 0, 0], [0, 1e20, 1], [7000, 0, 1], [0, 0, 1e9]], dtype=np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        t = _t_chain(Z["zc"], p)
        M3 = np.fmax(np.fmax(np.abs(p[:, 0]), np.abs(p[:, 1])), np.abs(p[:, 2])).astype(np.float32)        # v_max3_f32: a NaN operand is ignored
        assert not np.any(_sure(t, M3, Z["z_neg_k"], Z["z_h0"]))


def test_a_z_row_single_precision_cannot_serve_is_always_unsure():
    Z = ssd.prez_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, np.eye(3) * 1e9, np.zeros(3))
    assert Z["z_h0"] + Z["z_neg_k"] * 1e-3 < 0           # the bound follows the magnitude: any input beyond a millimetre is unsure
    Z = ssd.prez_host(-0.6, 0.6, 0.1, 1.3, -0.1, 1.1, np.eye(3), np.array([0.0, 0.0, 40000.0]))     # |c3| of four million bins: the bound exceeds a quarter bin
    assert Z["z_h0"] < 0


@pytest.mark.parametrize("seed", range(8))
def test_pixel_bound_on_random_calibrations(seed):
    rng = np.random.default_rng(3000 + seed)
    a = _rotation(rng)
    cam = rng.uniform(-1.0, 1.0, 3) * (2.0 if seed % 3 else 20.0)
    b = -a @ cam
    W, H = [(1024, 768), (1920, 1080), (640, 480), (1000, 750)][seed % 4]
    x_min, y_min = rng.uniform(-1.0, -0.3), rng.uniform(0.0, 0.4)
    x_max, y_max = x_min + rng.uniform(0.8, 2.0), y_min + rng.uniform(0.8, 2.0)
    Q = ssd.prexy_host(x_min, x_max, y_min, y_max, -0.1, 1.1, a, b)
    Z = ssd.prez_host(x_min, x_max, y_min, y_max, -0.1, 1.1, a, b, width=W, height=H)
    x_to_img, y_to_img = W / (x_max - x_min), H / (y_max - y_min)
    n = 60000
    k = n // 2
    # world points whose pixel coordinate lies on pixel edges +- small offsets (either axis), and anywhere in the range
    w = np.stack([rng.uniform(x_min, x_max, n), rng.uniform(y_min, y_max, n), rng.uniform(-0.1, 1.1, n)], 1)
    offs = rng.choice([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3, 3e-3, -3e-3], k) + rng.normal(0.0, 1e-5, k)
    axis = rng.integers(0, 2, k)
    ex, ey = rng.integers(0, W + 1, k), rng.integers(0, H + 1, k)
    w[:k, 0] = np.where(axis == 0, x_min + (ex + offs) / x_to_img, w[:k, 0])
    w[:k, 1] = np.where(axis == 1, y_max - (ey + offs) / y_to_img, w[:k, 1])
    p = ((w - b) @ np.linalg.inv(a).T).astype(np.float32)
    x, y, z = (p[:, i].astype(np.float64) for i in range(3))
    wx = ((a[0, 0] * x + a[0, 1] * y) + a[0, 2] * z) + b[0]
    wy = ((a[1, 0] * x + a[1, 1] * y) + a[1, 2] * z) + b[1]
    inr = (wx > x_min) & (wx < x_max) & (wy > y_min) & (wy < y_max)           # candidates are points in range
    ix = ((wx - x_min) * x_to_img).astype(np.int64)                          # Projection2D::worldToImage, pointcloud.cpp:79-83
    iy = ((y_max - wy) * y_to_img).astype(np.int64)
    dx, dy = _chain(Q["c"][:, 0], p), _chain(Q["c"][:, 1], p)
    for variant in range(2):
        if variant == 1:                                                    # a lane that went through the doubles carries d = fl32(D)
            dx = ((wx - x_min) * (256.0 / (x_max - x_min)) * 0.00390625 - 0.5).astype(np.float32)
            dy = ((wy - y_min) * (256.0 / (y_max - y_min)) * 0.00390625 - 0.5).astype(np.float32)
        px = _fma32(Z["f_w"], dx, np.full(len(p), Z["f_half_w"], np.float32))
        py = _fma32(Z["f_neg_h"], dy, np.full(len(p), Z["f_half_h"], np.float32))
        M3 = np.abs(p).max(1).astype(np.float32)
        gx = ((px - np.floor(px)).astype(np.float32) - np.float32(0.5)).astype(np.float32)
        gy = ((py - np.floor(py)).astype(np.float32) - np.float32(0.5)).astype(np.float32)
        hp = _fma32(Z["px_neg_k"], M3, np.full(len(p), Z["px_h0"], np.float32))
        sure = (np.maximum(np.abs(gx), np.abs(gy)) < hp) & inr
        assert np.array_equal(np.floor(px[sure]).astype(np.int64), ix[sure])
        assert np.array_equal(np.floor(py[sure]).astype(np.int64), iy[sure])
        assert ix[sure].min() >= 0 and ix[sure].max() < W and iy[sure].min() >= 0 and iy[sure].max() < H
        unsure_anywhere = 1.0 - sure[k:][inr[k:]].mean()
        assert 0 < (inr & ~sure).sum() and unsure_anywhere < (0.08 if np.abs(cam).max() > 3 else 0.02), unsure_anywhere
