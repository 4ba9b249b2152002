"""k_inquad's single-precision edge tests (csrc/ssd_quadtest.h: build_quad_edges, quad_edges_classify) on the CPU, against the REAL
reference's QuadrilateralTest (oracle/_ref, quadrilateralTest.cpp compiled where it lies) and the oracle's restatement of it.

Round 6: in the cells an edge of a quadrilateral runs through, k_inquad asks four single-precision half-planes first - on the d of
K1's range pre-filter - and the reference's test (bounding box, 3 x 3 cell map, up to two segments, doubles) only for the points
within the bound of an edge.  "Inside for sure" must mean that the reference says inside, "outside for sure" that it says
outside - and what the reference says is a matter of its map, which for long, thin, tilted quadrilaterals is NOT the geometry
(test_the_map_is_not_the_geometry...): build_quad_edges checks the map cell by cell and switches itself off where they differ."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ssd = importlib.import_module("stair-step-detector_amd")

X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX = -1.5, 1.5, 0.5, 3.5, -0.1, 1.9


def _calibration(rng, far=False):
    # a camera looking down at the stairs from 1 - 2 m (or from far off: large inputs), any roll
    ax, ay, az = rng.uniform(-0.9, -0.3), rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    a = rz @ ry @ rx
    cam = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-1.0, 0.3), rng.uniform(0.8, 2.0)]) * (20.0 if far else 1.0)
    return a, -a @ cam


def _tread(rng, yaw_max):
    # front-left, front-right, back-left, back-right of a tread seen from above: a rectangle turned by up to yaw_max, corners disturbed
    cx, cy = rng.uniform(-0.5, 0.5), rng.uniform(1.0, 3.0)
    w, h = rng.uniform(0.4, 1.2), rng.uniform(0.12, 0.4)
    yaw = rng.uniform(-yaw_max, yaw_max)
    c, s = np.cos(yaw), np.sin(yaw)
    q = np.array([[-w, -h], [w, -h], [-w, h], [w, h]]) * 0.5
    q = q @ np.array([[c, s], [-s, c]]) + [cx, cy] + rng.normal(0.0, 0.004, (4, 2))
    return q


def _ground(rng):
    # calcGroundQuadrilateral (pointcloud.cpp:489-512): the first step's front edge and its feet on y = yMin
    x0, x1 = rng.uniform(-0.9, -0.2), rng.uniform(0.2, 0.9)
    y0, y1 = rng.uniform(1.0, 2.0), 0.0
    y1 = y0 + rng.uniform(-0.25, 0.25)
    if y0 < y1:
        fl = [x0, Y_MIN]
        fr = [x1 + (y1 - Y_MIN) * (y1 - y0) / (x1 - x0), Y_MIN]
    else:
        fl = [x0 + (y0 - Y_MIN) * (y0 - y1) / (x0 - x1), Y_MIN]
        fr = [x1, Y_MIN]
    return np.array([fl, fr, [x0, y0], [x1, y1]])


def _any_convex(rng):
    ang = np.sort(rng.uniform(0.0, 2 * np.pi, 4))
    r = rng.uniform(0.15, 0.9, 4)
    p = np.stack([r * np.cos(ang), r * np.sin(ang) * rng.uniform(0.2, 1.0)], 1) + [rng.uniform(-0.5, 0.5), rng.uniform(1.2, 2.8)]
    return p[[0, 1, 3, 2]]           # cyclic 0 -> 1 -> 3 -> 2


def _points(rng, quad, a, b, n):
    """world points around a quadrilateral - anywhere in and about its box, and at chosen distances from each edge and corner -
    as the camera's floats"""
    lo, hi = quad.min(0) - 0.05, quad.max(0) + 0.05
    w = np.stack([rng.uniform(lo[0], hi[0], n), rng.uniform(lo[1], hi[1], n)], 1)
    order = [0, 1, 3, 2]
    k = n // 2
    e = rng.integers(0, 4, k)
    p0, p1 = quad[np.take(order, e)], quad[np.take(order, (e + 1) % 4)]
    t = rng.uniform(-0.1, 1.1, k)[:, None]
    t[: k // 8] = rng.choice([0.0, 1.0], (k // 8, 1))                      # at the corners
    along = p1 - p0
    normal = np.stack([-along[:, 1], along[:, 0]], 1) / np.linalg.norm(along, axis=1)[:, None]
    offs = rng.choice([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5, 3e-5, -3e-5, 1e-4, -1e-4, 1e-3, -1e-3], k)
    w[:k] = p0 + t * along + normal * (offs + rng.normal(0.0, 2e-7, k))[:, None]
    z = rng.uniform(0.0, 1.5, n)
    world = np.concatenate([w, z[:, None]], 1)
    return ((world - b) @ np.linalg.inv(a).T).astype(np.float32)


def _reference_answer(ref_or_oracle, quad, world_xy):
    rc, inside = ref_or_oracle.quad_test(quad.reshape(8), world_xy)
    return rc, inside.astype(bool)


def _check(ssd_mod, judge, quad, a, b, pts, stats):
    r = ssd_mod.quad_edges_host(quad.reshape(8), X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX, a, b, pts)
    rc, inside = _reference_answer(judge, quad, r["world_xy"])
    if r["err"] != 0 or rc != 0:
        assert (r["err"] != 0) == (rc != 0)
        assert np.isinf(r["m"])                              # the reference throws on this one: nothing is "for sure"
        stats["threw"] += 1
        return
    live = r["in_range_xy"].astype(bool)                     # k_inquad tests points in range only
    cls = r["cls"]
    assert not np.any(live & (cls > 0) & ~inside), "inside for sure, but the reference says outside"
    assert not np.any(live & (cls < 0) & inside), "outside for sure, but the reference says inside"
    stats["quads"] += 1
    stats["off"] += int(np.isinf(r["m"]))
    stats["points"] += int(live.sum())
    stats["unsure"] += int((live & (cls == 0)).sum())
    if np.isinf(r["m"]):
        assert not np.any(cls != 0)
    else:
        # farther than 0.1 mm from every edge's line nothing is left to the doubles (a range of 3 m: the band is some 10 um)
        assert 1e-7 < r["m"] < 1e-4
        assert abs(np.abs(r["gx"]) + np.abs(r["gy"]) - 1.0).max() < 1e-6


@pytest.mark.parametrize("kind,seed", [(k, s) for k in ("tread", "tread_turned", "ground", "convex") for s in range(6)])
def test_sure_answers_are_the_references(ssd, ref, kind, seed):
    rng = np.random.default_rng({"tread": 0, "tread_turned": 100, "ground": 200, "convex": 300}[kind] + seed)
    stats = dict(quads=0, off=0, points=0, unsure=0, threw=0)
    for it in range(60):
        a, b = _calibration(rng, far=(it % 7 == 6))
        quad = {"tread": lambda: _tread(rng, 0.2), "tread_turned": lambda: _tread(rng, 0.9), "ground": lambda: _ground(rng),
                "convex": lambda: _any_convex(rng)}[kind]()
        _check(ssd, ref, quad, a, b, _points(rng, quad, a, b, 3000), stats)
    assert stats["quads"] >= (20 if kind == "convex" else 50), stats       # (the reference throws on half of the arbitrary ones)
    if kind in ("tread", "ground"):
        # what a frame's quadrilaterals look like: the map agrees with the geometry, the single-precision test is in use
        assert stats["off"] == 0, stats
    # half of the points were put at up to 1 mm from an edge, most of them within 30 um: even so the doubles see a minority
    assert stats["unsure"] < 0.5 * stats["points"], stats


def test_sure_answers_are_the_oracles(ssd, oracle):
    # the same against the oracle's restatement (runs where oracle/_ref is not built)
    rng = np.random.default_rng(77)
    stats = dict(quads=0, off=0, points=0, unsure=0, threw=0)
    for it in range(80):
        a, b = _calibration(rng)
        quad = [_tread(rng, 0.3), _ground(rng), _any_convex(rng)][it % 3]
        _check(ssd, oracle, quad, a, b, _points(rng, quad, a, b, 2000), stats)
    assert stats["quads"] >= 70


def test_away_from_the_edges_nothing_is_left_to_the_doubles(ssd, oracle):
    rng = np.random.default_rng(5)
    a, b = _calibration(rng)
    quad = _tread(rng, 0.2)
    lo, hi = quad.min(0) - 0.2, quad.max(0) + 0.2
    n = 20000
    world = np.stack([rng.uniform(lo[0], hi[0], n), rng.uniform(lo[1], hi[1], n), rng.uniform(0.0, 1.0, n)], 1)
    pts = ((world - b) @ np.linalg.inv(a).T).astype(np.float32)
    r = ssd.quad_edges_host(quad.reshape(8), X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX, a, b, pts)
    assert np.isfinite(r["m"])
    live = r["in_range_xy"].astype(bool)
    assert live.sum() > n // 2
    assert (live & (r["cls"] == 0)).sum() <= 5               # a random point lies within 10 um of an edge once in thousands
    rc, inside = oracle.quad_test(quad.reshape(8), r["world_xy"])
    assert rc == 0 and 0.1 < inside[live].mean() < 0.9
    assert np.array_equal(r["cls"][live & (r["cls"] != 0)] > 0, inside.astype(bool)[live & (r["cls"] != 0)])


def test_the_map_is_not_the_geometry_for_a_thin_tilted_quadrilateral(ssd, oracle):
    """L -> U -> V -> R counterclockwise, convex.  The point P lies below the edge V -> R, outside the quadrilateral, in a cell of the
    reference's 3 x 3 map that only the box of the long edge L -> R meets: the reference tests that edge alone and says INSIDE
    (quadrilateralTest.cpp:318-372: a cell's segments are chosen by bounding boxes).  A half-plane test would say outside - so
    build_quad_edges must have switched itself off for this one."""
    L, U, V, R = [0.0, 0.7], [1.0, 0.5], [2.0, 1.5], [2.5, 3.5]
    quad = np.array([L, U, R, V]) + [-1.2, 0.0]                # 0 -> 1 -> 3 -> 2 = L -> U -> V -> R, inside the measuring range
    P = np.array([[2.2, 1.0]]) + [-1.2, 0.0]
    rc, inside = oracle.quad_test(quad.reshape(8), P)
    assert rc == 0 and inside[0] == 1                           # the reference's answer (the oracle is pinned to it: tests/test_oracle.py)
    # geometry: P is on the outer side of V -> R
    v, r = quad[3], quad[2]
    assert (r[0] - v[0]) * (P[0, 1] - v[1]) - (r[1] - v[1]) * (P[0, 0] - v[0]) < 0
    a, b = np.eye(3), np.zeros(3)
    pts = np.array([[P[0, 0], P[0, 1], 0.5]], dtype=np.float32)
    out = ssd.quad_edges_host(quad.reshape(8), X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX, a, b, pts)
    assert out["err"] == 0 and out["in_range_xy"][0] == 1
    assert np.isinf(out["m"]) and out["cls"][0] == 0


@pytest.mark.gpu
def test_the_device_builds_the_same_table_and_switches_the_same_quadrilaterals_off(ssd, gpu_device):
    """k_quads' three steps on the device - the coefficients, the check of the map's nine cells dealt out to lanes, the margin - give the host's
    table bit for bit (the CPU tests above hold the host's against the reference), for treads, ground trapezoids, arbitrary convex quadrilaterals and
    the thin tilted one: the same quadrilaterals lose the shortcut (m = infinity), the same ones the reference throws on."""
    rng = np.random.default_rng(2024)
    quads = [_tread(rng, 0.3) for _ in range(150)] + [_tread(rng, 0.9) for _ in range(100)] + [_ground(rng) for _ in range(100)] \
        + [_any_convex(rng) for _ in range(300)]
    quads.append(np.array([[0.0, 0.7], [1.0, 0.5], [2.5, 3.5], [2.0, 1.5]]) + [-1.2, 0.0])            # map and geometry differ
    dev = ssd.quad_edges_device(np.array(quads).reshape(-1, 8), X_MIN, X_MAX, Y_MIN, Y_MAX, device=gpu_device)
    a, b = np.eye(3), np.zeros(3)
    off = threw = 0
    for q, d in zip(quads, dev):
        h = ssd.quad_edges_host(q.reshape(8), X_MIN, X_MAX, Y_MIN, Y_MAX, Z_MIN, Z_MAX, a, b, np.zeros((1, 3), np.float32))
        host = np.concatenate([h["gx"], h["gy"], h["g2"], [h["m"]]]).astype(np.float32)
        if h["err"] != 0:
            threw += 1
            assert np.isinf(d[12])
            continue
        assert np.array_equal(host.view(np.uint32), d.view(np.uint32)), (q.tolist(), host.tolist(), d.tolist())
        off += int(np.isinf(d[12]))
    assert np.isinf(dev[-1][12])
    assert 30 < off < 300 and threw > 50, (off, threw)
